/* A libcsdr client written against the REFERENCE's own headers (libcsdr.h / libcsdr_gpl.h / fastddc.h), linked against
 * libcsdr_amd.so instead of libcsdr.so: the drop-in check of INTEGRATION.md section 1.  Prints results for the test. */
#include <stdio.h>
#include <stdlib.h>
#include "libcsdr.h"
#include "libcsdr_gpl.h"
#include "fastddc.h"
int main(void)
{
    enum { N = 16384 };
    complexf *in = malloc(sizeof(complexf) * N), *sh = malloc(sizeof(complexf) * N), *out = malloc(sizeof(complexf) * N);
    float *dem = malloc(sizeof(float) * N), *tmp = malloc(sizeof(float) * 4 * N);
    unsigned s = 12345;
    for (int i = 0; i < N; i++) { s = s * 1664525u + 1013904223u; in[i].i = (float)(s >> 8) / 8388608.0f - 1.0f; s = s * 1664525u + 1013904223u; in[i].q = (float)(s >> 8) / 8388608.0f - 1.0f; }
    float taps[79];
    int ntaps = firdes_filter_len(0.05f);
    firdes_lowpass_f(taps, ntaps, 0.5f / 10, WINDOW_HAMMING);
    shift_addition_data_t d = shift_addition_init(-0.085f);
    float phase = 0;
    for (int c = 0; c < N; c += 1024) phase = shift_addition_cc(in + c, sh + c, 1024, d, phase);
    int n = fir_decimate_cc(sh, out, N, 10, taps, ntaps);
    complexf last = {0, 0};
    last = fmdemod_quadri_cf(out, dem, n, tmp, last);
    double acc = 0; for (int i = 0; i < n; i++) acc += dem[i] * (double)dem[i];
    fastddc_t ddc; int err = fastddc_init(&ddc, 0.001f, 256, 0.0f);
    printf("ntaps=%d outputs=%d phase=%.6f demod_energy=%.6f last=(%.6f,%.6f) fft=%d err=%d\n", ntaps, n, phase, acc, last.i, last.q, ddc.fft_size, err);
    return 0;
}
