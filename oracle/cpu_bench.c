/* oracle/cpu_bench.c -- TEST/BENCH INFRASTRUCTURE (bench.py's cpu_baseline leg), not product code.
 *
 * Times the WFM receive chain (README.md:66) on host cores, in process, in the manner of the reference's own
 * micro-benchmark harness (test200.c:42-121: call the library on in-memory buffers, CLOCK_MONOTONIC_RAW):
 * one independent stream per thread, the CLI's block framing (16384-sample blocks, shift in 1024-chunks
 * csdr.c:911-918, FIR refeed csdr.c:1172-1174, fractional decimator refeed csdr.c:1517-1519).
 *
 *   -DUSE_REF : links oracle/_ref/libcsdr_ref.so = the unmodified reference ("kind": "reference");
 *               prototypes come from include/libcsdr_amd_compat.h, which mirrors the reference headers.
 *   default   : links liboracle.so, our C restatement ("kind": "port").
 *
 * usage: cpu_bench <threads> <seconds_of_signal_per_thread> ; prints one JSON object.
 */
#define _GNU_SOURCE
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>
#include <pthread.h>
#include <math.h>
#ifdef USE_REF
#include "../include/libcsdr_amd_compat.h"
#else
#include "csdr_oracle.h"
#endif

#define BLK 16384
typedef struct { int id; long n_blocks; double seconds; long audio; unsigned check; const float *taps; int ntaps; } job_t;

static double now(void) { struct timespec t; clock_gettime(CLOCK_MONOTONIC_RAW, &t); return t.tv_sec + t.tv_nsec * 1e-9; }

static void *worker(void *arg)
{
    job_t *j = (job_t *)arg;
    unsigned char *u8 = (unsigned char *)malloc(2 * BLK);
    unsigned seed = 42u + 977u * (unsigned)j->id;
    for (int k = 0; k < 2 * BLK; k++) { seed = seed * 1664525u + 1013904223u; u8[k] = (unsigned char)(seed >> 24); }
    float *xf = (float *)malloc(sizeof(float) * 2 * BLK);
    float *sh = (float *)malloc(sizeof(float) * 2 * BLK);
    float *firbuf = (float *)malloc(sizeof(float) * 2 * (2 * BLK));      /* leftover + new block */
    float *dec = (float *)malloc(sizeof(float) * 2 * (BLK / 5));
    float *dem = (float *)malloc(sizeof(float) * (BLK / 5));
    float *fdbuf = (float *)malloc(sizeof(float) * (BLK / 2));
    float *aud = (float *)malloc(sizeof(float) * BLK), *de = (float *)malloc(sizeof(float) * BLK);
    short *pcm = (short *)malloc(sizeof(short) * BLK);
    float *tmp = (float *)malloc(sizeof(float) * 4 * BLK);
    int fir_have = 0, fd_have = 0;
    float phase = 0, deemph_state = 0;
    long audio = 0; unsigned check = 0;
#ifdef USE_REF
    shift_addition_data_t sd = shift_addition_init(-0.085f);
    complexf last = {0, 0};
    fractional_decimator_ff_t fd = fractional_decimator_ff_init(5.0f, 12, NULL, 0);
#else
    orc_shift_addition_t sd = orc_shift_addition_init(-0.085f);
    orc_cf last = {0, 0};
    orc_fracdec_t fd; orc_fractional_decimator_ff_init(&fd, 5.0f, 12, NULL, 0);
#endif
    double t0 = now();
    for (long b = 0; b < j->n_blocks; b++) {
#ifdef USE_REF
        convert_u8_f(u8, xf, 2 * BLK);
        for (int c = 0; c < BLK; c += 1024) phase = shift_addition_cc((complexf *)xf + c, (complexf *)sh + c, 1024, sd, phase);
        memcpy(firbuf + 2 * fir_have, sh, sizeof(float) * 2 * BLK); fir_have += BLK;
        int nd = fir_decimate_cc((complexf *)firbuf, (complexf *)dec, fir_have, 10, (float *)j->taps, j->ntaps);
        memmove(firbuf, firbuf + 2 * 10 * nd, sizeof(float) * 2 * (fir_have - 10 * nd)); fir_have -= 10 * nd;
        last = fmdemod_quadri_cf((complexf *)dec, dem, nd, tmp, last);
        memcpy(fdbuf + fd_have, dem, sizeof(float) * nd); fd_have += nd;
        fractional_decimator_ff(fdbuf, aud, fd_have, &fd);
        memmove(fdbuf, fdbuf + fd.input_processed, sizeof(float) * (fd_have - fd.input_processed)); fd_have -= fd.input_processed;
        int na = fd.output_size;
        if (na > 0) { deemph_state = deemphasis_wfm_ff(aud, de, na, 50e-6f, 48000, deemph_state); convert_f_s16(de, pcm, na); }
#else
        orc_convert_u8_f(u8, xf, 2 * BLK);
        for (int c = 0; c < BLK; c += 1024) phase = orc_shift_addition_cc((orc_cf *)xf + c, (orc_cf *)sh + c, 1024, sd, phase);
        memcpy(firbuf + 2 * fir_have, sh, sizeof(float) * 2 * BLK); fir_have += BLK;
        int nd = orc_fir_decimate_cc((orc_cf *)firbuf, (orc_cf *)dec, fir_have, 10, j->taps, j->ntaps);
        memmove(firbuf, firbuf + 2 * 10 * nd, sizeof(float) * 2 * (fir_have - 10 * nd)); fir_have -= 10 * nd;
        last = orc_fmdemod_quadri_cf((orc_cf *)dec, dem, nd, last);
        memcpy(fdbuf + fd_have, dem, sizeof(float) * nd); fd_have += nd;
        orc_fractional_decimator_ff(fdbuf, aud, fd_have, &fd);
        memmove(fdbuf, fdbuf + fd.input_processed, sizeof(float) * (fd_have - fd.input_processed)); fd_have -= fd.input_processed;
        int na = fd.output_size;
        if (na > 0) { deemph_state = orc_deemphasis_wfm_ff(aud, de, na, 50e-6f, 48000, deemph_state); orc_convert_f_s16(de, pcm, na); }
#endif
        for (int k = 0; k < na; k++) check = check * 31u + (unsigned short)pcm[k];
        audio += na;
    }
    j->seconds = now() - t0; j->audio = audio; j->check = check;
    return NULL;
}

int main(int argc, char **argv)
{
    int threads = argc > 1 ? atoi(argv[1]) : 1;
    double sig_seconds = argc > 2 ? atof(argv[2]) : 10.0;
    long n_blocks = (long)(sig_seconds * 2.4e6 / BLK); if (n_blocks < 1) n_blocks = 1;
    float taps[79];
#ifdef USE_REF
    firdes_lowpass_f(taps, 79, 0.05f, WINDOW_HAMMING);
    const char *kind = "reference";
#else
    orc_firdes_lowpass_f(taps, 79, 0.05f, ORC_HAMMING);
    const char *kind = "port";
#endif
    pthread_t *th = (pthread_t *)malloc(sizeof(pthread_t) * threads);
    job_t *jobs = (job_t *)calloc(threads, sizeof(job_t));
    double t0 = now();
    for (int k = 0; k < threads; k++) { jobs[k].id = k; jobs[k].n_blocks = n_blocks; jobs[k].taps = taps; jobs[k].ntaps = 79; pthread_create(&th[k], NULL, worker, &jobs[k]); }
    double worst = 0; long audio = 0; unsigned check = 0;
    for (int k = 0; k < threads; k++) { pthread_join(th[k], NULL); if (jobs[k].seconds > worst) worst = jobs[k].seconds; audio += jobs[k].audio; check ^= jobs[k].check; }
    double wall = now() - t0;
    double samples = (double)threads * n_blocks * BLK;
    printf("{\"kind\": \"%s\", \"threads\": %d, \"samples\": %.0f, \"wall_s\": %.4f, \"slowest_thread_s\": %.4f, \"msps\": %.3f, \"audio_samples\": %ld, \"check\": %u}\n",
           kind, threads, samples, wall, worst, samples / wall / 1e6, audio, check);
    return 0;
}
