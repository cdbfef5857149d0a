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
 * usage: cpu_bench <threads> <seconds_of_signal_per_thread> [mode [p1 [p2]]] ; prints one JSON object.
 *   mode wfm (default)        config 2, README.md:66 chain, u8 IQ -> s16
 *   mode fir <D> <tbw>        config 1: fir_decimate_cc D tbw HAMMING on complexf with the CLI's 16384-sample blocks + refeed (csdr.c:1160-1176)
 *   mode fftfilt <taps>       config 3: apply_fir_fft_cc at fft_size 65536 with the CLI loop's buffers (csdr.c:1846-1880); amount = blocks per thread
 *   mode fastddc <channels>   config 4: fastddc_fwd_cc framing + FFT once per block, fastddc_inv_cc for the channels (D = 256, tbw = 0.001) spread over
 *                             the threads (thread t takes channels t, t+T, ...; every thread transforms the input block itself); amount = blocks
 *   mode nfm                  config 5: README.md:87 chain on one u8 IQ channel per thread
 * The modes other than wfm exist only in the -DUSE_REF build (they time the unmodified reference).  The FFT provider is whatever the
 * reference library was linked against: oracle/fftw_shim.c (double-precision radix 2, the checker's transform: cpu_bench_ref) or MKL's
 * FFTW3 interface (cpu_bench_ref_mkl: the fast CPU FFT; the baseline the benches quote for configs 3 and 4 when it is present).
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
typedef struct { int id, threads; long n_blocks; double seconds; long audio; unsigned check; const float *taps; int ntaps; int p1; float p2; double samples; } job_t;

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

#ifdef USE_REF
/* ---- config 1: fir_decimate_cc on complexf, CLI framing (csdr.c:1160-1176: the unconsumed tail is moved to the front and re-presented) */
static void *worker_fir(void *arg)
{
    job_t *j = (job_t *)arg;
    const int D = j->p1, nt = j->ntaps;
    float *in = (float *)malloc(sizeof(float) * 2 * BLK);
    unsigned seed = 1234u + 977u * (unsigned)j->id;
    for (int k = 0; k < 2 * BLK; k++) { seed = seed * 1664525u + 1013904223u; in[k] = (float)(seed >> 8) / 8388608.0f - 1.0f; }
    float *buf = (float *)malloc(sizeof(float) * 2 * (2 * BLK));
    float *out = (float *)malloc(sizeof(float) * 2 * (2 * BLK / D + 2));
    int have = 0; long produced = 0; unsigned check = 0;
    double t0 = now();
    for (long b = 0; b < j->n_blocks; b++) {
        memcpy(buf + 2 * have, in, sizeof(float) * 2 * BLK); have += BLK;
        int nd = fir_decimate_cc((complexf *)buf, (complexf *)out, have, D, (float *)j->taps, nt);
        memmove(buf, buf + 2 * D * nd, sizeof(float) * 2 * (have - D * nd)); have -= D * nd;
        produced += nd; if (nd) check = check * 31u + (unsigned)(out[0] * 1e6f);
    }
    j->seconds = now() - t0; j->audio = produced; j->check = check; j->samples = (double)j->n_blocks * BLK;
    return NULL;
}

/* ---- config 3: the bandpass_fir_fft_cc loop at a fixed fft_size of 65536 (csdr.c:1846-1880) */
static void *worker_fftfilt(void *arg)
{
    job_t *j = (job_t *)arg;
    const int fft = 65536, nt = j->p1, inp = fft - nt + 1, ovl = nt - 1;
    complexf *taps = (complexf *)calloc(sizeof(complexf), fft), *taps_fft = (complexf *)malloc(sizeof(complexf) * fft);
    FFT_PLAN_T *plan_taps = make_fft_c2c(fft, taps, taps_fft, 1, 0);
    complexf *input = (complexf *)calloc(sizeof(complexf), fft), *input_fourier = (complexf *)malloc(sizeof(complexf) * fft);
    FFT_PLAN_T *plan_forward = make_fft_c2c(fft, input, input_fourier, 1, 1);
    complexf *output_fourier = (complexf *)malloc(sizeof(complexf) * fft);
    complexf *o1 = (complexf *)calloc(sizeof(complexf), fft), *o2 = (complexf *)calloc(sizeof(complexf), fft);
    FFT_PLAN_T *pi1 = make_fft_c2c(fft, output_fourier, o1, 0, 1), *pi2 = make_fft_c2c(fft, output_fourier, o2, 0, 1);
    firdes_bandpass_c(taps, nt, -0.1f, 0.2f, WINDOW_HAMMING);
    fft_execute(plan_taps);
    complexf *src = (complexf *)malloc(sizeof(complexf) * inp), *dst = (complexf *)malloc(sizeof(complexf) * inp);
    unsigned seed = 3u + 977u * (unsigned)j->id;
    for (int k = 0; k < inp; k++) { seed = seed * 1664525u + 1013904223u; src[k].i = (float)(seed >> 8) / 8388608.0f - 1.0f; seed = seed * 1664525u + 1013904223u; src[k].q = (float)(seed >> 8) / 8388608.0f - 1.0f; }
    unsigned check = 0;
    double t0 = now();
    for (long b = 0; b < j->n_blocks; b++) {
        const int odd = (int)(b & 1);
        memcpy(input, src, sizeof(complexf) * inp);                                 /* fread(input, ..., input_size, stdin) */
        FFT_PLAN_T *pinv = odd ? pi2 : pi1, *plast = odd ? pi1 : pi2;
        complexf *last_overlap = (complexf *)plast->output + inp;
        apply_fir_fft_cc(plan_forward, pinv, taps_fft, last_overlap, ovl);
        memcpy(dst, pinv->output, sizeof(complexf) * inp);                          /* fwrite(plan_inverse->output, ...) */
        check = check * 31u + (unsigned)(dst[7].i * 1e3f);
    }
    j->seconds = now() - t0; j->audio = j->n_blocks * inp; j->check = check; j->samples = (double)j->n_blocks * inp;
    return NULL;
}

/* ---- config 4: fastddc_fwd_cc (csdr.c:2289-2299) + fastddc_inv_cc per channel (csdr.c:2363-2375, fastddc.c:106-166) */
static void *worker_fastddc(void *arg)
{
    job_t *j = (job_t *)arg;
    const int C = j->p1, T = j->threads, D = 256; const float tbw = j->p2;
    fastddc_t g; if (fastddc_init(&g, tbw, D, 0)) return NULL;
    const int fft = g.fft_size, inv = g.fft_inv_size;
    int mine = 0; for (int c = j->id; c < C; c += T) mine++;
    fastddc_t *ddc = (fastddc_t *)malloc(sizeof(fastddc_t) * (mine + 1));
    complexf **taps_fft = (complexf **)malloc(sizeof(complexf *) * (mine + 1));
    decimating_shift_addition_status_t *st = (decimating_shift_addition_status_t *)calloc(mine + 1, sizeof(*st));
    complexf *taps = (complexf *)calloc(sizeof(complexf), fft);
    int m = 0;
    for (int c = j->id; c < C; c += T, m++) {
        const float rate = -0.5f + ((float)c + 0.5f) / (float)C;
        fastddc_init(&ddc[m], tbw, D, rate);
        taps_fft[m] = (complexf *)malloc(sizeof(complexf) * fft);
        memset(taps, 0, sizeof(complexf) * fft);
        FFT_PLAN_T *pt = make_fft_c2c(fft, taps, taps_fft[m], 1, 0);
        const float hb = 0.5f / D;
        firdes_bandpass_c(taps, ddc[m].taps_length, (-rate) - hb, (-rate) + hb, WINDOW_HAMMING);
        fft_execute(pt); fft_swap_sides(taps_fft[m], fft); fft_destroy(pt);
    }
    complexf *input = (complexf *)calloc(sizeof(complexf), fft), *windowed = (complexf *)malloc(sizeof(complexf) * fft);
    complexf *spec = (complexf *)malloc(sizeof(complexf) * fft), *spec_w = (complexf *)malloc(sizeof(complexf) * fft);
    FFT_PLAN_T *plan = make_fft_c2c(fft, windowed, spec, 1, 1);
    complexf *inv_in = (complexf *)malloc(sizeof(complexf) * inv), *inv_out = (complexf *)malloc(sizeof(complexf) * inv);
    FFT_PLAN_T *plan_inverse = make_fft_c2c(inv, inv_in, inv_out, 0, 1);
    complexf *out = (complexf *)malloc(sizeof(complexf) * (g.post_input_size + 8));
    complexf *fresh = (complexf *)malloc(sizeof(complexf) * g.input_size);
    unsigned seed = 4u + 977u * (unsigned)j->id;
    for (int k = 0; k < g.input_size; k++) { seed = seed * 1664525u + 1013904223u; fresh[k].i = (float)(seed >> 8) / 8388608.0f - 1.0f; seed = seed * 1664525u + 1013904223u; fresh[k].q = (float)(seed >> 8) / 8388608.0f - 1.0f; }
    long produced = 0; unsigned check = 0;
    double t0 = now();
    for (long b = 0; b < j->n_blocks; b++) {
        for (int i = 0; i < g.overlap_length; i++) input[i] = input[i + g.input_size];
        memcpy(input + g.overlap_length, fresh, sizeof(complexf) * g.input_size);
        memcpy(windowed, input, sizeof(complexf) * fft);
        fft_execute(plan);
        for (int k = 0; k < mine; k++) {
            memcpy(spec_w, spec, sizeof(complexf) * fft);                           /* every fastddc_inv_cc process reads its own copy from the pipe; the function swaps it in place */
            st[k] = fastddc_inv_cc(spec_w, out, &ddc[k], plan_inverse, taps_fft[k], st[k]);
            produced += st[k].output_size; check = check * 31u + (unsigned)(out[0].i * 1e3f);
        }
    }
    j->seconds = now() - t0; j->audio = produced; j->check = check; j->samples = (double)j->n_blocks * g.input_size;
    return NULL;
}

/* ---- config 5: README.md:87, one u8 IQ channel per thread, the CLI's block framing (1024-sample audio-rate buffers) */
static void *worker_nfm(void *arg)
{
    job_t *j = (job_t *)arg;
    const int D = 50, nt = j->ntaps, AB = 1024;
    unsigned char *u8 = (unsigned char *)malloc(2 * BLK);
    unsigned seed = 5000u + 977u * (unsigned)j->id;
    for (int k = 0; k < 2 * BLK; k++) { seed = seed * 1664525u + 1013904223u; u8[k] = (unsigned char)(seed >> 24); }
    float *xf = (float *)malloc(sizeof(float) * 2 * BLK), *sh = (float *)malloc(sizeof(float) * 2 * BLK);
    float *firbuf = (float *)malloc(sizeof(float) * 2 * (2 * BLK)), *dec = (float *)malloc(sizeof(float) * 2 * (2 * BLK / D + 2));
    float *dem = (float *)malloc(sizeof(float) * (2 * BLK / D + 2)), *tmp = (float *)malloc(sizeof(float) * 4 * BLK);
    float *debuf = (float *)calloc(4 * AB + 2 * BLK / D, sizeof(float)), *de = (float *)malloc(sizeof(float) * (4 * AB + 2 * BLK / D));
    float *agcbuf = (float *)malloc(sizeof(float) * (4 * AB + 2 * BLK / D)), *agcout = (float *)malloc(sizeof(float) * AB);
    short *pcm = (short *)malloc(sizeof(short) * AB);
    fastagc_ff_t agc; memset(&agc, 0, sizeof(agc));
    agc.input_size = AB; agc.reference = 1.0f;
    agc.buffer_1 = (float *)calloc(AB, sizeof(float)); agc.buffer_2 = (float *)calloc(AB, sizeof(float)); agc.buffer_input = (float *)malloc(sizeof(float) * AB);
    shift_addition_data_t sd = shift_addition_init(-0.05f);
    complexf last = {0, 0};
    int fir_have = 0, de_have = AB /* csdr.c:1076-1081: the loop first filters the_bufsize zeros */, agc_have = 0;
    float phase = 0; long audio = 0; unsigned check = 0;
    double t0 = now();
    for (long b = 0; b < j->n_blocks; b++) {
        convert_u8_f(u8, xf, 2 * BLK);
        for (int c = 0; c < BLK; c += 1024) phase = shift_addition_cc((complexf *)xf + c, (complexf *)sh + c, 1024, sd, phase);
        memcpy(firbuf + 2 * fir_have, sh, sizeof(float) * 2 * BLK); fir_have += BLK;
        int nd = fir_decimate_cc((complexf *)firbuf, (complexf *)dec, fir_have, D, (float *)j->taps, nt);
        memmove(firbuf, firbuf + 2 * D * nd, sizeof(float) * 2 * (fir_have - D * nd)); fir_have -= D * nd;
        last = fmdemod_quadri_cf((complexf *)dec, dem, nd, tmp, last);
        limit_ff(dem, dem, nd, 1.0f);
        memcpy(debuf + de_have, dem, sizeof(float) * nd); de_have += nd;
        while (de_have >= AB) {                                                          /* csdr.c:1078-1084: the_bufsize-sample windows, tail re-fed */
            int got = deemphasis_nfm_ff(debuf, de, AB, 48000);
            memmove(debuf, debuf + got, sizeof(float) * (de_have - got)); de_have -= got;
            memcpy(agcbuf + agc_have, de, sizeof(float) * got); agc_have += got;
            while (agc_have >= AB) {
                memcpy(agc.buffer_input, agcbuf, sizeof(float) * AB);
                fastagc_ff(&agc, agcout);
                convert_f_s16(agcout, pcm, AB);
                memmove(agcbuf, agcbuf + AB, sizeof(float) * (agc_have - AB)); agc_have -= AB;
                audio += AB; check = check * 31u + (unsigned short)pcm[17];
            }
        }
    }
    j->seconds = now() - t0; j->audio = audio; j->check = check; j->samples = (double)j->n_blocks * BLK;
    return NULL;
}
#endif

int main(int argc, char **argv)
{
    int threads = argc > 1 ? atoi(argv[1]) : 1;
    double amount = argc > 2 ? atof(argv[2]) : 10.0;
    const char *mode = argc > 3 ? argv[3] : "wfm";
    long n_blocks = (long)(amount * 2.4e6 / BLK); if (n_blocks < 1) n_blocks = 1;
    static float taps[1025];
    int ntaps = 79, p1 = 0; float p2 = 0;
    void *(*fn)(void *) = worker;
    const char *what = "seconds of 2.4 MS/s signal per thread";
#ifdef USE_REF
    const char *kind = "reference";
    firdes_lowpass_f(taps, 79, 0.05f, WINDOW_HAMMING);
    if (!strcmp(mode, "fir")) {
        p1 = argc > 4 ? atoi(argv[4]) : 10; p2 = argc > 5 ? (float)atof(argv[5]) : 0.05f;
        ntaps = firdes_filter_len(p2); if (ntaps > 1025 || p1 < 1) { fprintf(stderr, "bad fir parameters\n"); return 2; }
        firdes_lowpass_f(taps, ntaps, 0.5f / (float)p1, WINDOW_HAMMING);                  /* csdr.c:1144-1158 */
        fn = worker_fir;
    } else if (!strcmp(mode, "fftfilt")) {
        p1 = argc > 4 ? atoi(argv[4]) : 1023; if (p1 < 1 || p1 > 65536) return 2;
        n_blocks = (long)amount; if (n_blocks < 1) n_blocks = 1; what = "blocks per thread";
        fn = worker_fftfilt;
    } else if (!strcmp(mode, "fastddc")) {
        p1 = argc > 4 ? atoi(argv[4]) : 256; p2 = argc > 5 ? (float)atof(argv[5]) : 0.001f;
        n_blocks = (long)amount; if (n_blocks < 1) n_blocks = 1; what = "blocks";
        fn = worker_fastddc;
    } else if (!strcmp(mode, "nfm")) {
        ntaps = firdes_filter_len(0.005f); firdes_lowpass_f(taps, ntaps, 0.5f / 50.0f, WINDOW_HAMMING);
        fn = worker_nfm;
    } else if (strcmp(mode, "wfm")) { fprintf(stderr, "unknown mode %s\n", mode); return 2; }
#else
    const char *kind = "port";
    orc_firdes_lowpass_f(taps, 79, 0.05f, ORC_HAMMING);
    if (strcmp(mode, "wfm")) { printf("{\"error\": \"mode %s needs the -DUSE_REF build (oracle/_ref/cpu_bench_ref)\"}\n", mode); return 3; }
#endif
    pthread_t *th = (pthread_t *)malloc(sizeof(pthread_t) * threads);
    job_t *jobs = (job_t *)calloc(threads, sizeof(job_t));
    double t0 = now();
    for (int k = 0; k < threads; k++) {
        jobs[k].id = k; jobs[k].threads = threads; jobs[k].n_blocks = n_blocks; jobs[k].taps = taps; jobs[k].ntaps = ntaps; jobs[k].p1 = p1; jobs[k].p2 = p2;
        jobs[k].samples = (double)n_blocks * BLK;
        pthread_create(&th[k], NULL, fn, &jobs[k]);
    }
    double worst = 0, busy = 0; long audio = 0; unsigned check = 0; double samples = 0;
    for (int k = 0; k < threads; k++) {
        pthread_join(th[k], NULL); if (jobs[k].seconds > worst) worst = jobs[k].seconds; busy += jobs[k].seconds;
        audio += jobs[k].audio; check ^= jobs[k].check; samples += jobs[k].samples;
    }
    double wall = now() - t0;
    /* fastddc: the threads share ONE wideband input (channels are split over them), so the job's input is one thread's sample count;
     * the rate is taken over the slowest thread's processing loop (filter design and FFT planning before the loop are set-up, not streaming) */
    double t_rate = wall;
    if (!strcmp(mode, "fastddc")) { samples = jobs[0].samples; t_rate = worst; }
    if (!strcmp(mode, "fftfilt")) t_rate = worst;
    printf("{\"kind\": \"%s\", \"mode\": \"%s\", \"threads\": %d, \"amount\": %g, \"amount_unit\": \"%s\", \"samples\": %.0f, \"wall_s\": %.4f, \"slowest_thread_s\": %.4f, "
           "\"thread_seconds\": %.4f, \"msps\": %.3f, \"audio_samples\": %ld, \"check\": %u}\n",
           kind, mode, threads, amount, what, samples, wall, worst, busy, samples / t_rate / 1e6, audio, check);
    return 0;
}
