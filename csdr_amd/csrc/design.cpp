// design.cpp -- host-side filter design and geometry for libcsdr_amd.so.
// These run once per filter (setup time), on the CPU, in the reference's own mixed float/double
// arithmetic so that taps and sizes agree with the reference (they define the operator the GPU applies).
#include "common.hpp"
#include <math.h>
#include <stdlib.h>
#include <string.h>
#include "nfm_deemph_taps.inc"

using namespace csdr_amd;

static float window_value(int window, float r)
{   // libcsdr.c:76-97: blackman / hamming / boxcar kernels; argument remapped to 0.5 + r/2
    if (window == CSDR_WINDOW_BOXCAR) return 1.0f;
    float x = (float)(0.5 + r / 2);
    if (window == CSDR_WINDOW_BLACKMAN) return (float)(0.42 - 0.5 * cos((double)(2 * PI_F * x)) + 0.08 * cos((double)(4 * PI_F * x)));   // C semantics: double cos of a float product
    return (float)(0.54 - 0.46 * cos((double)(2 * PI_F * x)));
}

extern "C" {

int csdr_amd_firdes_filter_len(float transition_bw)
{   // libcsdr.c:169-174
    int n = (int)(4.0 / transition_bw);
    return (n & 1) ? n : n + 1;
}

void csdr_amd_firdes_lowpass_f(float *taps, int length, float cutoff_rate, int window)
{   // libcsdr.c:127-142 + normalize_fir_f :117-125: symmetric windowed sinc with unit DC gain
    const int mid = length / 2;
    taps[mid] = 2 * PI_F * cutoff_rate * window_value(window, 0);
    for (int k = 1; k <= mid; k++) {
        const float arg = 2 * PI_F * cutoff_rate * k;
        const float v = (float)((sin((double)arg) / k) * window_value(window, (float)k / mid));
        taps[mid + k] = v; taps[mid - k] = v;
    }
    float dc = 0;
    for (int k = 0; k < length; k++) dc += taps[k];
    for (int k = 0; k < length; k++) taps[k] /= dc;
}

void csdr_amd_firdes_bandpass_c(csdr_complexf *taps, int length, float lowcut, float highcut, int window)
{   // libcsdr.c:144-167: low-pass prototype of half the band width, heterodyned to the band centre
    float *proto = (float *)malloc(sizeof(float) * (size_t)length);
    csdr_amd_firdes_lowpass_f(proto, length, (highcut - lowcut) / 2, window);
    const float centre = (highcut + lowcut) / 2;
    float ph = 0;
    for (int k = 0; k < length; k++) {
        const float c = (float)cos((double)ph), s = (float)sin((double)ph);
        ph += 2 * PI_F * centre;
        while (ph > 2 * PI_F) ph -= 2 * PI_F;
        while (ph < 0) ph += 2 * PI_F;
        taps[k].i = c * proto[k]; taps[k].q = s * proto[k];
    }
    free(proto);
}

void csdr_amd_shift_addition_init(float rate, float *out3)
{   // libcsdr_gpl.c:81-89
    rate *= 2;
    out3[0] = (float)sin((double)(rate * PI_F)); out3[1] = (float)cos((double)(rate * PI_F)); out3[2] = rate;
}

int csdr_amd_next_pow2(int x)
{   // libcsdr.c:1235-1243: strictly greater power of two
    for (int b = 0; b < 31; b++) if (x < (1 << b)) return 1 << b;
    return -1;
}

int csdr_amd_log2n(int x)
{   // libcsdr.c:1220-1233
    int at = -1;
    for (int b = 0; b < 31; b++) if ((x >> b) & 1) { if (at >= 0) return -1; at = b; }
    return at;
}

int csdr_amd_nfm_deemph_taps(int sample_rate, const float **taps)
{   // libcsdr.c:1113-1119 table selection; values from predefined.h:56-68 (see nfm_deemph_taps.inc)
    const uint32_t *bits = nullptr; int n = 0;
    switch (sample_rate) {
        case 48000: bits = nfm_deemph_bits_48000; n = (int)(sizeof(nfm_deemph_bits_48000) / 4); break;
        case 44100: bits = nfm_deemph_bits_44100; n = (int)(sizeof(nfm_deemph_bits_44100) / 4); break;
        case 8000:  bits = nfm_deemph_bits_8000;  n = (int)(sizeof(nfm_deemph_bits_8000) / 4); break;
        case 11025: bits = nfm_deemph_bits_11025; n = (int)(sizeof(nfm_deemph_bits_11025) / 4); break;
        default: if (taps) *taps = nullptr; return 0;
    }
    if (taps) *taps = (const float *)(const void *)bits;
    return n;
}

int csdr_amd_fastddc_init(csdr_fastddc_t *d, float transition_bw, int decimation, float shift_rate)
{   // fastddc.c:38-72
    d->pre_decimation = 1; d->post_decimation = decimation;
    // strip factors of two into the frequency-domain stage, leaving at least 2 for the time-domain stage
    while ((d->post_decimation % 2 == 0) && d->post_decimation / 2 != 1) { d->post_decimation /= 2; d->pre_decimation *= 2; }
    d->taps_min_length = csdr_amd_firdes_filter_len(transition_bw);
    d->taps_length = csdr_amd_next_pow2((int)(ceil(d->taps_min_length / (float)d->pre_decimation) * d->pre_decimation)) + 1;
    d->fft_size = csdr_amd_next_pow2(d->taps_length * 4);
    while (d->fft_size < d->pre_decimation) d->fft_size *= 2;
    d->overlap_length = d->taps_length - 1;
    d->input_size = d->fft_size - d->overlap_length;
    d->fft_inv_size = d->fft_size / d->pre_decimation;
    d->v = d->fft_size / d->overlap_length;
    const int middle = d->fft_size / 2;
    d->startbin = (int)(middle + middle * (-shift_rate) * 2);
    d->startbin = (int)(d->v * round(d->startbin / (float)d->v));
    d->offsetbin = d->startbin - middle;
    d->post_shift = d->pre_decimation * (shift_rate + ((float)d->offsetbin / d->fft_size));
    d->pre_shift = d->offsetbin / (float)d->fft_size;
    {   // decimating_shift_addition_init(post_shift, post_decimation): libcsdr_gpl.c:126-129 -> :81-89
        float r = d->post_shift * d->post_decimation; r *= 2;
        d->dsadata.sindelta = (float)sin((double)(r * PI_F));
        d->dsadata.cosdelta = (float)cos((double)(r * PI_F));
        d->dsadata.rate = r;
    }
    d->scrap = d->overlap_length / d->pre_decimation;
    d->post_input_size = d->fft_inv_size - d->scrap;
    d->output_scrape = 0;
    return d->fft_size <= 2;
}

} // extern "C"
