/* oracle/csdr_oracle.c -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.  See csdr_oracle.h.
 *
 * Plain-C restatement of the reference's algorithms for the hot path.  Each routine
 * states, in comments, the arithmetic types the reference's expression uses (float vs
 * double promotion matters for the bit-exact converters and for the float32 phase
 * bookkeeping of the shifters) and cites the reference file:line it follows.
 * Build: gcc -O2 -fno-fast-math -ffp-contract=off  (oracle/Makefile).
 */
#include "csdr_oracle.h"
#include "fftw3.h"
#include <math.h>
#include <stdlib.h>
#include <string.h>
#include <limits.h>

/* libcsdr.h:65  PI is a *float* constant in the reference; 2*PI etc. are float products. */
static const float PI_F = (float)3.14159265358979323846;

/* ------------------------------------------------------------------ design helpers */

int orc_firdes_filter_len(float transition_bw)
{   /* libcsdr.c:169-174: 4.0 (double) / float -> double, truncated to int, forced odd */
    int n = (int)(4.0 / transition_bw);
    return (n % 2 == 0) ? n + 1 : n;
}

static float window_kernel(int window, float r)
{   /* libcsdr.c:76-97.  "rate = 0.5 + rate/2" is computed in double and stored to float. */
    if (window == ORC_BOXCAR) return 1.0f;
    float x = (float)(0.5 + r / 2);
    if (window == ORC_BLACKMAN)
        return (float)(0.42 - 0.5 * cos(2 * PI_F * x) + 0.08 * cos(4 * PI_F * x));
    return (float)(0.54 - 0.46 * cos(2 * PI_F * x));   /* HAMMING is also the default (libcsdr.h:75) */
}

void orc_firdes_lowpass_f(float *taps, int length, float cutoff_rate, int window)
{   /* libcsdr.c:127-142: windowed sinc around the middle tap, then unit-DC-gain normalisation
     * (libcsdr.c:117-125).  sin() argument is a float product; the quotient and the window product
     * are double; each tap is rounded to float once. */
    int mid = length / 2;
    taps[mid] = 2 * PI_F * cutoff_rate * window_kernel(window, 0);
    for (int k = 1; k <= mid; k++) {
        float arg = 2 * PI_F * cutoff_rate * k;
        float v = (float)((sin((double)arg) / k) * window_kernel(window, (float)k / mid));
        taps[mid - k] = v; taps[mid + k] = v;
    }
    float sum = 0;
    for (int k = 0; k < length; k++) sum += taps[k];
    for (int k = 0; k < length; k++) taps[k] = taps[k] / sum;
}

void orc_firdes_bandpass_c(orc_cf *taps, int length, float lowcut, float highcut, int window)
{   /* libcsdr.c:144-167: real low-pass of half the pass-band width, modulated to the band centre;
     * the modulation phase is a float accumulator wrapped into [0, 2*pi]. */
    float *lp = (float *)malloc(sizeof(float) * length);
    orc_firdes_lowpass_f(lp, length, (highcut - lowcut) / 2, window);
    float centre = (highcut + lowcut) / 2, phase = 0;
    for (int k = 0; k < length; k++) {
        float c = (float)cos((double)phase), s = (float)sin((double)phase);
        phase += 2 * PI_F * centre;
        while (phase > 2 * PI_F) phase -= 2 * PI_F;
        while (phase < 0) phase += 2 * PI_F;
        taps[k].i = c * lp[k]; taps[k].q = s * lp[k];
    }
    free(lp);
}

int orc_next_pow2(int x)
{   /* libcsdr.c:1235-1243: smallest power of two STRICTLY greater than x */
    for (int b = 0; b < 31; b++) if (x < (1 << b)) return 1 << b;
    return -1;
}

int orc_log2n(int x)
{   /* libcsdr.c:1220-1233: exponent if x is a power of two, else -1 */
    int found = -1;
    for (int b = 0; b < 31; b++) if ((x >> b) & 1) { if (found != -1) return -1; found = b; }
    return found;
}

/* ------------------------------------------------------------------ converters */

/* x86 cvttss2si / cvttsd2si semantics: out-of-range and NaN give INT_MIN ("integer indefinite"). */
static int trunc_f_to_i32(float x)  { return (x >= -2147483648.0f && x < 2147483648.0f) ? (int)x : INT_MIN; }
static int trunc_d_to_i32(double x) { return (x > -2147483649.0 && x < 2147483648.0) ? (int)x : INT_MIN; }

void orc_convert_u8_f(const unsigned char *in, float *out, int n)
{   /* libcsdr.c:2365: (float)v / (255/2.0) - 1.0, evaluated in double, stored as float */
    for (int k = 0; k < n; k++) out[k] = (float)((double)(float)in[k] / (UCHAR_MAX / 2.0) - 1.0);
}
void orc_convert_s8_f(const signed char *in, float *out, int n)
{   /* libcsdr.c:2370: "/SCHAR_MAX" in float.  The reference is built with -ffast-math (Makefile:38), whose
     * -freciprocal-math turns the division into a multiplication by the rounded reciprocal; the shipped
     * binary's values are the contract (16 of 256 codes differ from a true division). */
    const float r = 1.0f / (float)SCHAR_MAX;
    for (int k = 0; k < n; k++) out[k] = (float)in[k] * r;
}
void orc_convert_s16_f(const short *in, float *out, int n)
{   /* libcsdr.c:2375: "/SHRT_MAX" in float -> reciprocal multiplication under the reference's
     * -ffast-math build (1536 of 65536 codes differ from a true division) */
    const float r = 1.0f / (float)SHRT_MAX;
    for (int k = 0; k < n; k++) out[k] = (float)in[k] * r;
}
void orc_convert_f_u8(const float *in, unsigned char *out, int n)
{   /* libcsdr.c:2380: x*255 in float, *0.5 + 128 in double, truncated, narrowed modulo 256 */
    for (int k = 0; k < n; k++) out[k] = (unsigned char)trunc_d_to_i32((double)(in[k] * (float)UCHAR_MAX) * 0.5 + 128);
}
void orc_convert_f_s8(const float *in, signed char *out, int n)
{   /* libcsdr.c:2387 */
    for (int k = 0; k < n; k++) out[k] = (signed char)trunc_f_to_i32(in[k] * (float)SCHAR_MAX);
}
void orc_convert_f_s16(const float *in, short *out, int n)
{   /* libcsdr.c:2397: x*32767 in float, truncated toward zero, narrowed modulo 2^16 (no clipping) */
    for (int k = 0; k < n; k++) out[k] = (short)trunc_f_to_i32(in[k] * (float)SHRT_MAX);
}
void orc_convert_f_s24(const float *in, unsigned char *out, int n, int bigendian)
{   /* libcsdr.c:2403-2423: int32 = trunc(x * 8388607); flag set -> bytes LSB first, flag clear -> MSB first
     * (the flag name is inverted in the reference on a little-endian host; we keep the behaviour). */
    for (int k = 0; k < n; k++) {
        unsigned v = (unsigned)trunc_f_to_i32(in[k] * (float)(INT_MAX >> 8));
        unsigned char b0 = v & 0xff, b1 = (v >> 8) & 0xff, b2 = (v >> 16) & 0xff;
        if (bigendian) { out[3 * k] = b0; out[3 * k + 1] = b1; out[3 * k + 2] = b2; }
        else           { out[3 * k] = b2; out[3 * k + 1] = b1; out[3 * k + 2] = b0; }
    }
}
void orc_convert_s24_f(const unsigned char *in, float *out, int n, int bigendian)
{   /* libcsdr.c:2425-2437: the three bytes go to the top of an int32, divided by (float)(INT_MAX-256) */
    for (int k = 0; k < n; k++) {
        const unsigned char *p = in + 3 * k;
        unsigned u = bigendian ? ((unsigned)p[2] << 24) | ((unsigned)p[1] << 16) | ((unsigned)p[0] << 8)
                               : ((unsigned)p[2] << 8) | ((unsigned)p[1] << 16) | ((unsigned)p[0] << 24);
        out[k] = (float)(int)u * (1.0f / (float)(INT_MAX - 256));   /* reciprocal form, see convert_s8_f */
    }
}

/* ------------------------------------------------------------------ shifters */

static inline orc_cf rot(orc_cf x, float c, float s)
{   /* (I + jQ)(c + js), four float products, two float sums (e.g. libcsdr.c:199-200) */
    orc_cf y; y.i = c * x.i - s * x.q; y.q = s * x.i + c * x.q; return y;
}

float orc_shift_math_cc(const orc_cf *in, orc_cf *out, int n, float rate, float starting_phase)
{   /* libcsdr.c:186-207: libm cos/sin of a float phase that is advanced and wrapped to [0,2pi] PER SAMPLE */
    float phase = starting_phase, inc = (rate * 2) * PI_F;
    for (int k = 0; k < n; k++) {
        out[k] = rot(in[k], (float)cos((double)phase), (float)sin((double)phase));
        phase += inc;
        while (phase > 2 * PI_F) phase -= 2 * PI_F;
        while (phase < 0) phase += 2 * PI_F;
    }
    return phase;
}

void orc_shift_table_init(float *table, int table_size)
{   /* libcsdr.c:211-222: quarter-wave sine */
    for (int k = 0; k < table_size; k++) table[k] = (float)sin((double)(((float)k / table_size) * (PI_F / 2)));
}

float orc_shift_table_cc(const orc_cf *in, orc_cf *out, int n, float rate, const float *table, int table_size, float starting_phase)
{   /* libcsdr.c:229-265: quadrant folding into the quarter-wave table with index truncation */
    float phase = starting_phase, inc = (rate * 2) * PI_F, q90 = PI_F / 2;
    for (int k = 0; k < n; k++) {
        int quadrant = (int)(phase / q90);
        float within = phase - quadrant * q90;
        int si = (int)((within / q90) * table_size), ci = table_size - 1 - si;
        if (quadrant & 1) { int t = si; si = ci; ci = t; }
        if (si < 0) si = 0; if (si >= table_size) si = table_size - 1;   /* the reference would read out of bounds here */
        if (ci < 0) ci = 0; if (ci >= table_size) ci = table_size - 1;
        float s = ((quadrant > 1) ? -1 : 1) * table[si];
        float c = ((quadrant && quadrant < 3) ? -1 : 1) * table[ci];
        out[k] = rot(in[k], c, s);
        phase += inc;
        while (phase > 2 * PI_F) phase -= 2 * PI_F;
        while (phase < 0) phase += 2 * PI_F;
    }
    return phase;
}

static float wrap_pm_pi(float p)
{
    while (p > PI_F) p -= 2 * PI_F;
    while (p < -PI_F) p += 2 * PI_F;
    return p;
}

float orc_shift_unroll_init(float rate, int size, float *dsin, float *dcos)
{   /* libcsdr.c:268-284: table entry k holds the angle of (k+1) increments, accumulated in float */
    float inc = 2 * rate * PI_F, acc = 0;
    for (int k = 0; k < size; k++) {
        acc = wrap_pm_pi(acc + inc);
        dsin[k] = (float)sin((double)acc); dcos[k] = (float)cos((double)acc);
    }
    return inc;
}

float orc_shift_unroll_cc(const orc_cf *in, orc_cf *out, int n, const float *dsin, const float *dcos, float phase_increment, float starting_phase)
{   /* libcsdr.c:286-305 */
    float c0 = (float)cos((double)starting_phase), s0 = (float)sin((double)starting_phase);
    for (int k = 0; k < n; k++) {
        float c = c0 * dcos[k] - s0 * dsin[k], s = s0 * dcos[k] + c0 * dsin[k];
        out[k] = rot(in[k], c, s);
    }
    return wrap_pm_pi(starting_phase + n * phase_increment);
}

float orc_shift_addfast_init(float rate, float *dsin4, float *dcos4)
{   /* libcsdr.c:307-317 */
    float inc = 2 * rate * PI_F;
    for (int j = 0; j < 4; j++) { dsin4[j] = (float)sin((double)(inc * (j + 1))); dcos4[j] = (float)cos((double)(inc * (j + 1))); }
    return inc;
}

float orc_shift_addfast_cc(const orc_cf *in, orc_cf *out, int n, const float *dsin4, const float *dcos4, float phase_increment, float starting_phase)
{   /* libcsdr.c:406-434 (the C, non-NEON body): groups of four advance from the previous group's 4th phasor */
    float c0 = (float)cos((double)starting_phase), s0 = (float)sin((double)starting_phase);
    for (int g = 0; g < n / 4; g++) {
        float c[4], s[4];
        for (int j = 0; j < 4; j++) { c[j] = c0 * dcos4[j] - s0 * dsin4[j]; s[j] = s0 * dcos4[j] + c0 * dsin4[j]; }
        for (int j = 0; j < 4; j++) out[4 * g + j] = rot(in[4 * g + j], c[j], s[j]);
        c0 = c[3]; s0 = s[3];
    }
    return wrap_pm_pi(starting_phase + n * phase_increment);
}

orc_shift_addition_t orc_shift_addition_init(float rate)
{   /* libcsdr_gpl.c:81-89: the stored rate is already doubled */
    orc_shift_addition_t d; rate *= 2;
    d.sindelta = (float)sin((double)(rate * PI_F)); d.cosdelta = (float)cos((double)(rate * PI_F)); d.rate = rate;
    return d;
}

float orc_shift_addition_cc(const orc_cf *in, orc_cf *out, int n, orc_shift_addition_t d, float starting_phase)
{   /* libcsdr_gpl.c:27-52: phasor recurrence in float, re-seeded from libm at every call */
    float c = (float)cos((double)starting_phase), s = (float)sin((double)starting_phase);
    for (int k = 0; k < n; k++) {
        out[k] = rot(in[k], c, s);
        float c1 = c * d.cosdelta - s * d.sindelta, s1 = s * d.cosdelta + c * d.sindelta;
        c = c1; s = s1;
    }
    return wrap_pm_pi(starting_phase + d.rate * PI_F * n);
}

float orc_shift_addition_fc(const float *in, orc_cf *out, int n, orc_shift_addition_t d, float starting_phase)
{   /* libcsdr_gpl.c:54-79: real input */
    float c = (float)cos((double)starting_phase), s = (float)sin((double)starting_phase);
    for (int k = 0; k < n; k++) {
        out[k].i = c * in[k]; out[k].q = s * in[k];
        float c1 = c * d.cosdelta - s * d.sindelta, s1 = s * d.cosdelta + c * d.sindelta;
        c = c1; s = s1;
    }
    return wrap_pm_pi(starting_phase + d.rate * PI_F * n);
}

orc_shift_addition_t orc_decimating_shift_addition_init(float rate, int decimation)
{ return orc_shift_addition_init(rate * decimation); }   /* libcsdr_gpl.c:126-129 */

orc_dsa_status_t orc_decimating_shift_addition_cc(const orc_cf *in, orc_cf *out, int n, orc_shift_addition_t d, int decimation, orc_dsa_status_t st)
{   /* libcsdr_gpl.c:131-160: rotate every decimation-th sample starting at decimation_remain */
    float c = (float)cos((double)st.starting_phase), s = (float)sin((double)st.starting_phase);
    int pos, produced = 0;
    for (pos = st.decimation_remain; pos < n; pos += decimation) {
        out[produced++] = rot(in[pos], c, s);
        float c1 = c * d.cosdelta - s * d.sindelta, s1 = s * d.cosdelta + c * d.sindelta;
        c = c1; s = s1;
    }
    st.decimation_remain = pos - n;
    st.starting_phase = wrap_pm_pi(st.starting_phase + d.rate * PI_F * produced);
    st.output_size = produced;
    return st;
}

/* ------------------------------------------------------------------ filters, demod, audio */

int orc_fir_decimate_cc(const orc_cf *in, orc_cf *out, int n, int decimation, const float *taps, int taps_length)
{   /* libcsdr.c:528-549: real taps, I and Q accumulated separately from t = 0 upward, float */
    int produced = 0;
    for (int base = 0; base < n && base + taps_length <= n; base += decimation) {
        float ai = 0, aq = 0;
        for (int t = 0; t < taps_length; t++) ai += in[base + t].i * taps[t];
        for (int t = 0; t < taps_length; t++) aq += in[base + t].q * taps[t];
        out[produced].i = ai; out[produced].q = aq; produced++;
    }
    return produced;
}

orc_cf orc_fmdemod_quadri_cf(const orc_cf *in, float *out, int n, orc_cf last)
{   /* libcsdr.c:1021,1040-1071: K*(I dQ - Q dI)/(I^2+Q^2); numerator/denominator float, K is a double
     * literal so the scaling and the division are double, rounded to float once; 0 when the power is 0 */
    const double K = 0.340447550238101026565118445432744920253753662109375;
    for (int k = 0; k < n; k++) {
        orc_cf prev = k ? in[k - 1] : last;
        float dq = in[k].q - prev.q, di = in[k].i - prev.i;
        float num = in[k].i * dq - in[k].q * di;
        float den = in[k].i * in[k].i + in[k].q * in[k].q;
        out[k] = den ? (float)(K * num / den) : 0;
    }
    return in[n - 1];
}

float orc_deemphasis_wfm_ff(const float *in, float *out, int n, float tau, int sample_rate, float last_output)
{   /* libcsdr.c:1081-1097: one-pole low-pass, all float; NaN state is reset to 0 */
    float dt = (float)(1.0 / sample_rate), alpha = dt / (tau + dt);
    if (last_output != last_output) last_output = 0;
    for (int k = 0; k < n; k++) { last_output = alpha * in[k] + (1 - alpha) * last_output; out[k] = last_output; }
    return last_output;
}

int orc_deemphasis_nfm_ff(const float *in, float *out, int n, const float *taps, int taps_length)
{   /* libcsdr.c:1101-1128: fixed FIR (tables predefined.h:56-68), outputs for i < n - taps_length */
    if (!taps_length) return 0;
    int k;
    for (k = 0; k < n - taps_length; k++) {
        float acc = 0;
        for (int t = 0; t < taps_length; t++) acc += taps[t] * in[k + t];
        out[k] = acc;
    }
    return k;
}

void orc_limit_ff(const float *in, float *out, int n, float m)
{   /* libcsdr.c:1130-1137 */
    for (int k = 0; k < n; k++) { float v = (m < in[k]) ? m : in[k]; out[k] = (-m > v) ? -m : v; }
}

void orc_gain_ff(const float *in, float *out, int n, float g)
{ for (int k = 0; k < n; k++) out[k] = g * in[k]; }   /* libcsdr.c:1139-1142 */

void orc_fastagc_ff(orc_fastagc_t *st, float *out)
{   /* libcsdr.c:946-991: three-block look-ahead AGC; gain ramps linearly across the block being emitted
     * (the block received two calls ago); ramp arithmetic is double (1.0-rate) */
    int n = st->input_size;
    float peak_in = 0;
    for (int k = 0; k < n; k++) { float a = fabsf(st->buffer_input[k]); if (a > peak_in) peak_in = a; }
    float peak = peak_in;
    if (peak < st->peak_2) peak = st->peak_2;
    if (peak < st->peak_1) peak = st->peak_1;
    float target = st->reference / peak;
    if (target > 50) target = 50;                          /* FASTAGC_MAX_GAIN, libcsdr.c:944 */
    for (int k = 0; k < n; k++) {
        float r = (float)k / n;
        float g = (float)(st->last_gain * (1.0 - r) + target * r);
        out[k] = st->buffer_1[k] * g;
    }
    float *recycled = st->buffer_1;
    st->buffer_1 = st->buffer_2; st->peak_1 = st->peak_2;
    st->buffer_2 = st->buffer_input; st->peak_2 = peak_in;
    st->buffer_input = recycled; st->last_gain = target;
}

void orc_fractional_decimator_ff_init(orc_fracdec_t *d, float rate, int num_poly_points, const float *taps, int taps_length)
{   /* libcsdr.c:715-748 */
    memset(d, 0, sizeof(*d));
    d->num_poly_points = num_poly_points & ~1;
    d->xifirst = -(num_poly_points / 2) + 1; d->xilast = num_poly_points / 2;
    int idx = 0;
    for (int a = d->xifirst; a <= d->xilast; a++, idx++) {
        float prod = 1;
        for (int b = d->xifirst; b <= d->xilast; b++) if (a != b) prod *= (a - b);
        d->denom[idx] = prod;
    }
    d->where = -d->xifirst; d->rate = rate; d->taps = taps; d->taps_length = taps_length; d->input_processed = 0;
}

void orc_fractional_decimator_ff(const float *in, float *out, int n, orc_fracdec_t *d)
{   /* libcsdr.c:751-793: Lagrange interpolation over num_poly_points samples around the float position `where` */
    int produced = 0, hi, P = d->num_poly_points;
    float y[64], coef[64];
    for (; (hi = (int)ceilf(d->where)) + P + d->taps_length < n; d->where += d->rate) {
        int lo = hi - 1;
        for (int w = 0; w < P; w++) {
            if (d->taps) { float acc = 0; for (int t = 0; t < d->taps_length; t++) acc += d->taps[t] * in[lo + w + t]; y[w] = acc; }
            else y[w] = in[lo + w];
        }
        float x = d->where - lo;
        int idx = 0;
        for (int a = d->xifirst; a <= d->xilast; a++, idx++) {
            float prod = 1;
            for (int b = d->xifirst; b <= d->xilast; b++) if (a != b) prod *= (x - b);
            coef[idx] = prod;
        }
        float acc = 0;
        for (int w = 0; w < P; w++) acc += (coef[w] / d->denom[w]) * y[w];
        out[produced++] = acc;
    }
    d->input_processed = (hi - 1) + d->xifirst;
    d->where -= d->input_processed;
    d->output_size = produced;
}

/* ------------------------------------------------------------------ FFT paths */

void orc_fft_c2c(const orc_cf *in, orc_cf *out, int n, int forward)
{   /* fft_fftw.c:6-15,36-39 -> unnormalised DFT, sign -1 forward / +1 backward */
    fftwf_plan p = fftwf_plan_dft_1d(n, (fftwf_complex *)in, (fftwf_complex *)out, forward ? FFTW_FORWARD : FFTW_BACKWARD, FFTW_ESTIMATE);
    fftwf_execute(p); fftwf_destroy_plan(p);
}

void orc_apply_fir_fft_cc(const orc_cf *in, orc_cf *result, int fft_size, const orc_cf *taps_fft, const orc_cf *last_overlap, int overlap)
{   /* libcsdr.c:814-849: FFT -> bin-wise complex product -> IFFT -> /N on every bin -> add the saved overlap */
    orc_cf *spec = (orc_cf *)malloc(sizeof(orc_cf) * fft_size), *prod = (orc_cf *)malloc(sizeof(orc_cf) * fft_size);
    orc_fft_c2c(in, spec, fft_size, 1);
    for (int k = 0; k < fft_size; k++) {
        prod[k].i = spec[k].i * taps_fft[k].i - spec[k].q * taps_fft[k].q;
        prod[k].q = spec[k].i * taps_fft[k].q + spec[k].q * taps_fft[k].i;
    }
    orc_fft_c2c(prod, result, fft_size, 0);
    for (int k = 0; k < fft_size; k++) { result[k].i /= fft_size; result[k].q /= fft_size; }
    for (int k = 0; k < overlap; k++) { result[k].i += last_overlap[k].i; result[k].q += last_overlap[k].q; }
    free(spec); free(prod);
}

int orc_fastddc_init(orc_fastddc_t *ddc, float transition_bw, int decimation, float shift_rate)
{   /* fastddc.c:38-72: split D into a power-of-two frequency-domain part and a small time-domain part,
     * derive the FFT geometry, quantise the coarse shift to a multiple of v bins */
    ddc->pre_decimation = 1; ddc->post_decimation = decimation;
    while (floorf((float)ddc->post_decimation / 2) == (float)ddc->post_decimation / 2 && ddc->post_decimation / 2 != 1) {
        ddc->post_decimation /= 2; ddc->pre_decimation *= 2;
    }
    ddc->taps_min_length = orc_firdes_filter_len(transition_bw);
    ddc->taps_length = orc_next_pow2((int)(ceil(ddc->taps_min_length / (float)ddc->pre_decimation) * ddc->pre_decimation)) + 1;
    ddc->fft_size = orc_next_pow2(ddc->taps_length * 4);
    while (ddc->fft_size < ddc->pre_decimation) ddc->fft_size *= 2;
    ddc->overlap_length = ddc->taps_length - 1;
    ddc->input_size = ddc->fft_size - ddc->overlap_length;
    ddc->fft_inv_size = ddc->fft_size / ddc->pre_decimation;
    ddc->v = ddc->fft_size / ddc->overlap_length;
    int middle = ddc->fft_size / 2;
    ddc->startbin = (int)(middle + middle * (-shift_rate) * 2);          /* float expression truncated to int */
    ddc->startbin = (int)(ddc->v * round(ddc->startbin / (float)ddc->v));
    ddc->offsetbin = ddc->startbin - middle;
    ddc->post_shift = ddc->pre_decimation * (shift_rate + ((float)ddc->offsetbin / ddc->fft_size));
    ddc->pre_shift = ddc->offsetbin / (float)ddc->fft_size;
    ddc->dsadata = orc_decimating_shift_addition_init(ddc->post_shift, ddc->post_decimation);
    ddc->scrap = ddc->overlap_length / ddc->pre_decimation;
    ddc->post_input_size = ddc->fft_inv_size - ddc->scrap;
    ddc->output_scrape = 0;
    return ddc->fft_size <= 2;
}

void orc_fft_swap_sides(orc_cf *io, int fft_size)
{   /* fastddc.c:91-104: exchange the two halves (fftshift for even sizes) */
    int h = fft_size / 2;
    for (int k = 0; k < h; k++) { orc_cf t = io[k]; io[k] = io[k + h]; io[k + h] = t; }
}

orc_dsa_status_t orc_fastddc_inv_cc(const orc_cf *spectrum, orc_cf *out, const orc_fastddc_t *ddc, const orc_cf *taps_fft, orc_dsa_status_t st)
{   /* fastddc.c:106-166: fftshift the big spectrum, fold all fft_size filtered bins into fft_inv_size
     * bins (alias = decimate by pre_decimation), scale, fftshift back, small inverse FFT, /fft_inv_size,
     * drop `scrap` leading samples, then the residual fine shift + post-decimation. */
    int N = ddc->fft_size, M = ddc->fft_inv_size;
    orc_cf *sh = (orc_cf *)malloc(sizeof(orc_cf) * N);
    memcpy(sh, spectrum, sizeof(orc_cf) * N); orc_fft_swap_sides(sh, N);
    orc_cf *acc = (orc_cf *)calloc(M, sizeof(orc_cf)), *td = (orc_cf *)malloc(sizeof(orc_cf) * M);
    for (int k = 0; k < N; k++) {
        int dst = (N + k - ddc->offsetbin + M / 2) % M;
        acc[dst].i += sh[k].i * taps_fft[k].i - sh[k].q * taps_fft[k].q;
        acc[dst].q += sh[k].i * taps_fft[k].q + sh[k].q * taps_fft[k].i;
    }
    for (int k = 0; k < M; k++) { acc[k].i /= ddc->pre_decimation; acc[k].q /= ddc->pre_decimation; }
    orc_fft_swap_sides(acc, M);
    orc_fft_c2c(acc, td, M, 0);
    for (int k = 0; k < M; k++) { td[k].i /= M; td[k].q /= M; }
    st = orc_decimating_shift_addition_cc(td + ddc->scrap, out, ddc->post_input_size, ddc->dsadata, ddc->post_decimation, st);
    free(sh); free(acc); free(td);
    return st;
}

/* ------------------------------------------------------------------ whole-stream CLI models */

float orc_stream_shift_addition_cc(const orc_cf *in, orc_cf *out, long n, float rate, float starting_phase, int chunk)
{   /* csdr.c:896-923: shift_addition_cc is applied in chunks of 1024 samples, phase carried in float */
    orc_shift_addition_t d = orc_shift_addition_init(rate);
    for (long pos = 0; pos < n; pos += chunk) {
        int len = (n - pos > chunk) ? chunk : (int)(n - pos);
        starting_phase = orc_shift_addition_cc(in + pos, out + pos, len, d, starting_phase);
    }
    return starting_phase;
}

long orc_stream_fir_decimate_cc(const orc_cf *in, orc_cf *out, long n, int decimation, const float *taps, int taps_length)
{   /* csdr.c:1160-1176: the refeed rule (unconsumed tail re-presented) makes the block loop equal to one
     * long convolution sampled every `decimation` inputs, with no zero history in front */
    long produced = 0;
    for (long base = 0; base + taps_length <= n; base += decimation) {
        float ai = 0, aq = 0;
        for (int t = 0; t < taps_length; t++) ai += in[base + t].i * taps[t];
        for (int t = 0; t < taps_length; t++) aq += in[base + t].q * taps[t];
        out[produced].i = ai; out[produced].q = aq; produced++;
    }
    return produced;
}

long orc_stream_wfm_chain(const unsigned char *iq_u8, long n_complex, float shift_rate, int decimation,
                          const float *taps, int taps_length, int frac_rate, float tau, int audio_rate,
                          short *audio_s16, float *audio_f)
{   /* README.md:66 / csdr-fm:41, stage by stage on one stream:
     *   convert_u8_f | shift_addition_cc r | fir_decimate_cc D | fmdemod_quadri_cf |
     *   fractional_decimator_ff R (integer R, no prefilter: csdr.c:1465-1525 -> libcsdr.c:751-793) |
     *   deemphasis_wfm_ff | convert_f_s16 */
    float *xf = (float *)malloc(sizeof(float) * 2 * n_complex);
    orc_convert_u8_f(iq_u8, xf, (int)(2 * n_complex));
    orc_cf *sh = (orc_cf *)malloc(sizeof(orc_cf) * n_complex);
    orc_stream_shift_addition_cc((orc_cf *)xf, sh, n_complex, shift_rate, 0.0f, 1024);
    free(xf);
    orc_cf *dec = (orc_cf *)malloc(sizeof(orc_cf) * (n_complex / decimation + 1));
    long nd = orc_stream_fir_decimate_cc(sh, dec, n_complex, decimation, taps, taps_length);
    free(sh);
    if (nd <= 0) { free(dec); return 0; }
    float *dem = (float *)malloc(sizeof(float) * nd);
    orc_cf zero = {0, 0};
    orc_fmdemod_quadri_cf(dec, dem, (int)nd, zero);
    free(dec);
    /* fractional decimator run as ONE block over the stream (its streaming loop is block-size invariant
     * for the produced samples; only the tail differs) */
    orc_fracdec_t fd; orc_fractional_decimator_ff_init(&fd, (float)frac_rate, 12, NULL, 0);
    float *aud = (float *)malloc(sizeof(float) * (nd / frac_rate + 2));
    orc_fractional_decimator_ff(dem, aud, (int)nd, &fd);
    long na = fd.output_size;
    free(dem);
    float *de = (float *)malloc(sizeof(float) * (na + 1));
    orc_deemphasis_wfm_ff(aud, de, (int)na, tau, audio_rate, 0.0f);
    if (audio_f) memcpy(audio_f, de, sizeof(float) * na);
    if (audio_s16) orc_convert_f_s16(de, audio_s16, (int)na);
    free(aud); free(de);
    return na;
}


/* ================================================================== f2 blocks */
void orc_amdemod_cf(const orc_cf *in, float *out, int n)
{   /* libcsdr.c:861-873: i*i+q*q in float, then sqrt() -- the double sqrt of a float, rounded back to float */
    for (int k = 0; k < n; k++) out[k] = in[k].i * in[k].i + in[k].q * in[k].q;
    for (int k = 0; k < n; k++) out[k] = (float)sqrt((double)out[k]);
}

void orc_amdemod_estimator_cf(const orc_cf *in, float *out, int n, float alpha, float beta)
{   /* libcsdr.c:875-901: alpha*max(|i|,|q|) + beta*min(|i|,|q|); (0,*) selects the minimum-RMS-error pair (double literals -> float) */
    if (alpha == 0) { alpha = 0.947543636291; beta = 0.392485425092; }
    for (int k = 0; k < n; k++) {
        float ai = in[k].i; if (ai < 0) ai = -ai;
        float aq = in[k].q; if (aq < 0) aq = -aq;
        float mx = ai; if (aq > mx) mx = aq;
        float mn = ai; if (aq < mn) mn = aq;
        out[k] = alpha * mx + beta * mn;
    }
}

float orc_fmdemod_atan_cf(const orc_cf *in, float *out, int n, float last_phase)
{   /* libcsdr.c:1004-1019: phase = (float)atan2(q, i) (double atan2, libcsdr.h:52); unwrap with the FLOAT constant PI (libcsdr.h:65) */
    const float PIf = (float)3.14159265358979323846;
    for (int k = 0; k < n; k++) {
        float phase = (float)atan2((double)in[k].q, (double)in[k].i);
        float d = phase - last_phase;
        if (d < -PIf) d += 2 * PIf;
        if (d > PIf) d -= 2 * PIf;
        out[k] = d / PIf;
        last_phase = phase;
    }
    return last_phase;
}

orc_dcblock_t orc_dcblock_ff(const float *in, float *out, int n, float a, orc_dcblock_t p)
{   /* libcsdr.c:903-918: y[i] = x[i] - x[i-1] + a*y[i-1]; a == 0 selects 0.999 */
    if (a == 0) a = 0.999;
    out[0] = in[0] - p.last_input + a * p.last_output;
    for (int k = 1; k < n; k++) out[k] = in[k] - in[k - 1] + a * out[k - 1];
    p.last_input = in[n - 1]; p.last_output = out[n - 1];
    return p;
}

float orc_fastdcblock_ff(const float *in, float *out, int n, float last_dc_level)
{   /* libcsdr.c:920-941: block mean, removal level ramps linearly from the previous block's mean to this one's */
    float avg = 0.0f;
    for (int k = 0; k < n; k++) avg += in[k];
    avg /= n;
    const float diff = avg - last_dc_level;
    for (int k = 0; k < n; k++) { float lvl = last_dc_level + diff * ((float)k / n); out[k] = in[k] - lvl; }
    return avg;
}

float orc_agc_ff(const float *in, float *out, int n, float reference, float attack_rate, float decay_rate, float max_gain,
                 short hang_time, short attack_wait_time, float gain_filter_alpha, float last_gain)
{   /* libcsdr_gpl.c:163-260: envelope-following AGC with hang / attack-wait counters (reset every call) and a one-pole filter on the gain */
    short hang_counter = 0, attack_wait_counter = 0;
    float gain = last_gain, last_peak = reference / last_gain, dgain;
    out[0] = last_gain * in[0];
    for (int k = 1; k < n; k++) {
        const float a = fabsf(in[k]);
        const float error = reference / a - gain;
        if (in[k] != 0) {
            if (error < 0) {
                if (last_peak < a) { attack_wait_counter = attack_wait_time; last_peak = a; }
                if (attack_wait_counter > 0) { attack_wait_counter--; dgain = 0; }
                else { dgain = error * attack_rate; hang_counter = hang_time; }
            } else {
                if (hang_counter > 0) { hang_counter--; dgain = 0; }
                else dgain = error * decay_rate;
            }
            gain = gain + dgain;
        }
        if (gain > max_gain) gain = max_gain;
        if (gain < 0) gain = 0;
        gain = gain + last_gain - gain_filter_alpha * last_gain;
        out[k] = gain * in[k];
        last_gain = gain;
    }
    return gain;
}

void orc_realpart_cf(const orc_cf *in, float *out, int n) { for (int k = 0; k < n; k++) out[k] = in[k].i; }   /* csdr.c:634-645 */

void orc_logpower_cf(const orc_cf *in, float *out, int n, float add_db)
{   /* libcsdr.c:1296-1303: |x|^2 in float, log10() in double rounded to float, then 10*y + add_db in float */
    for (int k = 0; k < n; k++) out[k] = in[k].i * in[k].i + in[k].q * in[k].q;
    for (int k = 0; k < n; k++) out[k] = (float)log10((double)out[k]);
    for (int k = 0; k < n; k++) out[k] = 10 * out[k] + add_db;
}

void orc_precalculate_window(float *windowt, int size, int window)
{   /* libcsdr.c:1256-1267: kernel(2*rate + 1) with rate = (float)i/(size-1); the argument is formed in double and passed as float */
    for (int k = 0; k < size; k++) { float rate = (float)k / (size - 1); windowt[k] = window_kernel(window, (float)(2.0 * rate + 1.0)); }
}

void orc_apply_precalculated_window_c(const orc_cf *in, orc_cf *out, int size, const float *windowt)
{ for (int k = 0; k < size; k++) { out[k].i = in[k].i * windowt[k]; out[k].q = in[k].q * windowt[k]; } }   /* libcsdr.c:1269-1276 */


/* ================================================================== f3: IMA ADPCM (ima_adpcm.c:88-174) */
static const int adpcm_index_adjust[16] = { -1, -1, -1, -1, 2, 4, 6, 8, -1, -1, -1, -1, 2, 4, 6, 8 };          /* ima_adpcm.c:90-95 */
static int adpcm_step(int index)
{   /* ima_adpcm.c:98-108: the standard IMA table, 89 entries, 7 ... 32767 */
    static const int t[89] = { 7, 8, 9, 10, 11, 12, 13, 14, 16, 17, 19, 21, 23, 25, 28, 31, 34, 37, 41, 45, 50, 55, 60, 66, 73, 80, 88, 97, 107, 118, 130, 143,
        157, 173, 190, 209, 230, 253, 279, 307, 337, 371, 408, 449, 494, 544, 598, 658, 724, 796, 876, 963, 1060, 1166, 1282, 1411, 1552, 1707, 1878, 2066,
        2272, 2499, 2749, 3024, 3327, 3660, 4026, 4428, 4871, 5358, 5894, 6484, 7132, 7845, 8630, 9493, 10442, 11487, 12635, 13899, 15289, 16818, 18500,
        20350, 22385, 24623, 27086, 29794, 32767 };
    return t[index];
}
static short adpcm_decode_one(unsigned code, orc_adpcm_t *st)
{   /* ima_adpcm.c:110-134 */
    const int step = adpcm_step(st->index);
    int diff = step >> 3;
    if (code & 1) diff += step >> 2;
    if (code & 2) diff += step >> 1;
    if (code & 4) diff += step;
    if (code & 8) diff = -diff;
    st->previousValue += diff;
    if (st->previousValue > 32767) st->previousValue = 32767; else if (st->previousValue < -32768) st->previousValue = -32768;
    st->index += adpcm_index_adjust[code];
    if (st->index < 0) st->index = 0; else if (st->index > 88) st->index = 88;
    return (short)st->previousValue;
}
static unsigned adpcm_encode_one(short sample, orc_adpcm_t *st)
{   /* ima_adpcm.c:136-152 */
    int diff = sample - st->previousValue, step = adpcm_step(st->index);
    unsigned code = 0;
    if (diff < 0) { code = 8; diff = -diff; }
    if (diff >= step) { code |= 4; diff -= step; }
    step >>= 1;
    if (diff >= step) { code |= 2; diff -= step; }
    step >>= 1;
    if (diff >= step) { code |= 1; }
    adpcm_decode_one(code, st);
    return code;
}
orc_adpcm_t orc_encode_ima_adpcm_i16_u8(const short *in, unsigned char *out, int n, orc_adpcm_t st)
{   /* ima_adpcm.c:154-163: two samples per byte, low nibble first; an odd last sample is dropped */
    for (int k = 0; k < n / 2; k++) { unsigned lo = adpcm_encode_one(in[2 * k], &st), hi = adpcm_encode_one(in[2 * k + 1], &st); out[k] = (unsigned char)(lo | (hi << 4)); }
    return st;
}
orc_adpcm_t orc_decode_ima_adpcm_u8_i16(const unsigned char *in, short *out, int n, orc_adpcm_t st)
{   /* ima_adpcm.c:165-174 */
    for (int k = 0; k < n; k++) { out[2 * k] = adpcm_decode_one(in[k] & 0xf, &st); out[2 * k + 1] = adpcm_decode_one((in[k] >> 4) & 0xf, &st); }
    return st;
}
void orc_compress_fft_adpcm_f_u8(const float *in, unsigned char *out, int fft_size)
{   /* csdr.c:1745-1768: the first value repeated 10 times in front, (short)(v*100), encoder restarted from the zero state for every block */
    short tmp[fft_size + 10];
    for (int k = 0; k < fft_size + 10; k++) { float v = (k < 10 ? in[0] : in[k - 10]) * 100; tmp[k] = (short)(int)v; }
    orc_adpcm_t st = {0, 0};
    orc_encode_ima_adpcm_i16_u8(tmp, out, fft_size + 10, st);
}
