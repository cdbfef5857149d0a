/* include/csdr_amd.h -- C ABI of the MI355X-native libcsdr hot path (libcsdr_amd.so).
 *
 * Two layers, both plain C (no torch / C++ types in any signature):
 *
 *  1. DEVICE BATCH API  (csdr_amd_*):  device pointers, N independent sample streams per call,
 *     stream-major layout  buf[stream * pitch + sample]  (pitch in ELEMENTS of the buffer's type),
 *     asynchronous on the context's HIP stream.  This is what bench.py and the parity tests drive.
 *     Cross-block state (phases, last samples, filter history) is explicit and caller-owned, exactly
 *     like the reference's by-value state, but held in device arrays of
 *     n_streams entries so that a call never synchronises with the host.
 *
 *  2. DROP-IN HOST API  (include/libcsdr_amd_compat.h): the reference's own symbols
 *     (libcsdr.h:85-229, libcsdr_gpl.h:26-46, fastddc.h:26-29, fft_fftw.h:24-27) with identical
 *     signatures and struct layouts, operating on host pointers (H2D -> kernels -> D2H).
 *
 * Every entry point cites the reference interface it replaces.  Return value: 0 on success,
 * negative on error (csdr_amd_last_error() gives the text) unless stated otherwise.
 * The library FAILS LOUDLY (error return, never a CPU fallback) when no gfx950 device is usable.
 */
#ifndef CSDR_AMD_H
#define CSDR_AMD_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* libcsdr.h:46 -- interleaved I/Q pair, 8 bytes */
typedef struct csdr_complexf_s { float i, q; } csdr_complexf;

/* libcsdr.h:70-73 */
typedef enum { CSDR_WINDOW_BOXCAR = 0, CSDR_WINDOW_BLACKMAN = 1, CSDR_WINDOW_HAMMING = 2 } csdr_window_t;

typedef struct csdr_amd_ctx csdr_amd_ctx;

/* ------------------------------------------------------------------ context / memory */
/* hip_stream: a hipStream_t owned by the caller (e.g. torch.cuda.current_stream().cuda_stream) or
 * NULL to let the context create its own non-blocking stream. */
csdr_amd_ctx *csdr_amd_ctx_create(int device, void *hip_stream);
void          csdr_amd_ctx_destroy(csdr_amd_ctx *ctx);
int           csdr_amd_ctx_sync(csdr_amd_ctx *ctx);
void         *csdr_amd_ctx_stream(csdr_amd_ctx *ctx);
const char   *csdr_amd_last_error(void);
int           csdr_amd_device_count(void);
/* "gfx950:sramecc+:xnack-" style architecture name of the context's device */
const char   *csdr_amd_device_arch(csdr_amd_ctx *ctx);

void *csdr_amd_malloc(csdr_amd_ctx *ctx, size_t bytes);
void  csdr_amd_free(csdr_amd_ctx *ctx, void *dptr);
int   csdr_amd_h2d(csdr_amd_ctx *ctx, void *dst_dev, const void *src_host, size_t bytes);   /* sync */
int   csdr_amd_d2h(csdr_amd_ctx *ctx, void *dst_host, const void *src_dev, size_t bytes);   /* sync */
int   csdr_amd_d2d(csdr_amd_ctx *ctx, void *dst_dev, const void *src_dev, size_t bytes);    /* async, regions must not overlap */
int   csdr_amd_memset(csdr_amd_ctx *ctx, void *dst_dev, int value, size_t bytes);           /* async */

/* HIP-event timing on the context's stream (bench.py: the kernel stream is not torch's stream) */
int   csdr_amd_timer_start(csdr_amd_ctx *ctx);
int   csdr_amd_timer_stop_ms(csdr_amd_ctx *ctx, float *ms);   /* records, synchronises, returns elapsed */

/* ------------------------------------------------------------------ host-side design helpers */
int  csdr_amd_firdes_filter_len(float transition_bw);                                        /* libcsdr.c:169-174 */
void csdr_amd_firdes_lowpass_f(float *taps, int length, float cutoff_rate, int window);      /* libcsdr.c:117-142 */
void csdr_amd_firdes_bandpass_c(csdr_complexf *taps, int length, float lowcut, float highcut, int window); /* :144-167 */
int  csdr_amd_next_pow2(int x);                                                              /* libcsdr.c:1235-1243 */
int  csdr_amd_log2n(int x);                                                                  /* libcsdr.c:1220-1233 */
/* fixed NFM de-emphasis FIR tables (predefined.h:56-68): returns tap count or 0 for an unsupported rate */
int  csdr_amd_nfm_deemph_taps(int sample_rate, const float **taps);

/* ------------------------------------------------------------------ sample-format converters (bit exact)
 * libcsdr.c:2363-2437 / libcsdr.h:220-229.  n = number of VALUES (not bytes). */
int csdr_amd_convert_u8_f (csdr_amd_ctx *ctx, const uint8_t *in, float *out, size_t n);
int csdr_amd_convert_s8_f (csdr_amd_ctx *ctx, const int8_t *in, float *out, size_t n);
int csdr_amd_convert_s16_f(csdr_amd_ctx *ctx, const int16_t *in, float *out, size_t n);
int csdr_amd_convert_f_u8 (csdr_amd_ctx *ctx, const float *in, uint8_t *out, size_t n);
int csdr_amd_convert_f_s8 (csdr_amd_ctx *ctx, const float *in, int8_t *out, size_t n);
int csdr_amd_convert_f_s16(csdr_amd_ctx *ctx, const float *in, int16_t *out, size_t n);
int csdr_amd_convert_f_s24(csdr_amd_ctx *ctx, const float *in, uint8_t *out, size_t n, int bigendian);
int csdr_amd_convert_s24_f(csdr_amd_ctx *ctx, const uint8_t *in, float *out, size_t n, int bigendian);

/* ------------------------------------------------------------------ frequency shifters
 * All shifter variants are "rotator generators" (the variant's own float32 phase bookkeeping replayed
 * exactly on the device) feeding one mixing kernel.  The rotator table rot[0..n) is shared by every
 * stream that has the same rate and starting phase.
 *
 * phase_io: HOST float; read as the starting phase, overwritten with the phase after n samples (the reference
 * returns it by value: libcsdr_gpl.c:48-51, libcsdr.c:206, 302-304).  The float32 phase recurrences are strictly
 * sequential and data independent; they run on the host (identical IEEE arithmetic) and are uploaded, the
 * device does the parallel part (sin/cos, per-chunk phasor replay, mixing).  Calls do not block on the GPU. */
enum {
    CSDR_SHIFT_ADDITION = 0,   /* shift_addition_cc  libcsdr_gpl.c:27-52, 1024-chunks per csdr.c:911-918 */
    CSDR_SHIFT_MATH     = 1,   /* shift_math_cc      libcsdr.c:186-207 */
    CSDR_SHIFT_TABLE    = 2,   /* shift_table_cc     libcsdr.c:229-265 (aux = table_size) */
    CSDR_SHIFT_UNROLL   = 3,   /* shift_unroll_cc    libcsdr.c:268-305 (aux = table size = chunk, csdr.c:821) */
    CSDR_SHIFT_ADDFAST  = 4    /* shift_addfast_cc   libcsdr.c:307-317, 406-434, 1024-chunks csdr.c:785 */
};
int csdr_amd_rotator_generate(csdr_amd_ctx *ctx, int variant, float rate, float *phase_io,
                              csdr_complexf *rot, size_t n, int chunk, int aux);
/* out[s][k] = in[s][k] * rot[k]   (four float products, two sums, no FMA contraction) */
int csdr_amd_mix_cc(csdr_amd_ctx *ctx, const csdr_complexf *in, csdr_complexf *out, const csdr_complexf *rot,
                    int n_streams, size_t n, size_t in_pitch, size_t out_pitch);
/* real input variant: shift_addition_fc libcsdr_gpl.c:54-79 */
int csdr_amd_mix_fc(csdr_amd_ctx *ctx, const float *in, csdr_complexf *out, const csdr_complexf *rot,
                    int n_streams, size_t n, size_t in_pitch, size_t out_pitch);
/* convenience: generate + mix with context scratch */
int csdr_amd_shift_cc(csdr_amd_ctx *ctx, int variant, float rate, float *phase_io,
                      const csdr_complexf *in, csdr_complexf *out, int n_streams, size_t n,
                      size_t in_pitch, size_t out_pitch, int chunk, int aux);

/* decimating_shift_addition_cc (libcsdr_gpl.c:131-160), one call per stream-block.
 * dsa_data: device array of n_streams {sindelta, cosdelta, rate} = shift_addition_data_t (libcsdr_gpl.h:26-31) as
 * returned by decimating_shift_addition_init (csdr_amd_shift_addition_init(rate*decimation) on the host).
 * status_io: device int32[3*n_streams] = {decimation_remain, float bits of starting_phase, output_size}
 * laid out exactly like decimating_shift_addition_status_t (libcsdr_gpl.h:39-44). */
int csdr_amd_decimating_shift_addition_cc(csdr_amd_ctx *ctx, const csdr_complexf *in, csdr_complexf *out,
                                          int n_streams, int input_size, size_t in_pitch, size_t out_pitch,
                                          const void *dsa_data, int decimation, void *status_io);
/* shift_addition_init libcsdr_gpl.c:81-89 (host): out3 = {sindelta, cosdelta, 2*rate} */
void csdr_amd_shift_addition_init(float rate, float *out3);

/* ------------------------------------------------------------------ FIR decimator
 * fir_decimate_cc libcsdr.c:528-549: out[s][o] = sum_t in[s][D*o+t]*taps[t] for all o with D*o+taps<=input_size.
 * Returns the number of outputs per stream (>=0) or a negative error.  taps: DEVICE pointer. */
/* the kernel the calling thread's last csdr_amd_fir_decimate_cc launched: k_fir_poly (short filters), k_fir_mfma3 / k_fir_mfma (long filters on the fp32 matrix cores),
 * k_fir_generic */
const char *csdr_amd_fir_last_kernel(void);
int csdr_amd_fir_decimate_cc(csdr_amd_ctx *ctx, const csdr_complexf *in, csdr_complexf *out,
                             int n_streams, int input_size, size_t in_pitch, size_t out_pitch,
                             int decimation, const float *taps, int taps_length);
/* real FIR without decimation = deemphasis_nfm_ff libcsdr.c:1101-1128 (outputs for i < input_size-taps_length).
 * Returns outputs per stream. */
int csdr_amd_fir_ff(csdr_amd_ctx *ctx, const float *in, float *out, int n_streams, int input_size,
                    size_t in_pitch, size_t out_pitch, const float *taps, int taps_length);

/* ------------------------------------------------------------------ demod / audio
 * fmdemod_quadri_cf libcsdr.c:1040-1071.  last_io: device complexf[n_streams] (in: previous block's last
 * sample, out: this block's last sample). */
int csdr_amd_fmdemod_quadri_cf(csdr_amd_ctx *ctx, const csdr_complexf *in, float *out, int n_streams,
                               size_t n, size_t in_pitch, size_t out_pitch, csdr_complexf *last_io);
/* limit_ff / gain_ff libcsdr.c:1130-1142 (flat arrays) */
int csdr_amd_limit_ff(csdr_amd_ctx *ctx, const float *in, float *out, size_t n, float max_amplitude);
int csdr_amd_gain_ff(csdr_amd_ctx *ctx, const float *in, float *out, size_t n, float gain);
/* deemphasis_wfm_ff libcsdr.c:1081-1097.  last_io: device float[n_streams]. */
int csdr_amd_deemphasis_wfm_ff(csdr_amd_ctx *ctx, const float *in, float *out, int n_streams, size_t n,
                               size_t in_pitch, size_t out_pitch, float tau, int sample_rate, float *last_io);
/* fastagc_ff libcsdr.c:946-991 over n_blocks consecutive blocks per stream.
 * state_io: device float[n_streams * (2*block + 4)] = {buffer_1[block], buffer_2[block], peak_1, peak_2,
 * last_gain, pad}; zero-initialised = the CLI's calloc'ed state (csdr.c:1393-1394). */
int csdr_amd_fastagc_ff(csdr_amd_ctx *ctx, const float *in, float *out, int n_streams, int n_blocks, int block,
                        size_t in_pitch, size_t out_pitch, float reference, float *state_io);

/* fractional_decimator_ff libcsdr.c:715-793 (Lagrange, optional FIR prefilter).  The float `where`
 * bookkeeping is data independent, so the plan (sample indices + interpolation coefficients) is built on the
 * host with the reference's float arithmetic and applied to all streams by one kernel.
 * state: opaque host object carrying where/input_processed between calls. */
typedef struct csdr_amd_fracdec csdr_amd_fracdec;
csdr_amd_fracdec *csdr_amd_fracdec_create(float rate, int num_poly_points, const float *host_taps, int taps_length);
void csdr_amd_fracdec_destroy(csdr_amd_fracdec *d);
/* returns outputs per stream; *input_processed receives the reference's d->input_processed */
int csdr_amd_fractional_decimator_ff(csdr_amd_ctx *ctx, csdr_amd_fracdec *d, const float *in, float *out,
                                     int n_streams, int input_size, size_t in_pitch, size_t out_pitch,
                                     int *input_processed);
void  csdr_amd_fracdec_set_where(csdr_amd_fracdec *d, float where);   /* fractional_decimator_ff_t.where */
/* the_bufsize > 0: csdr_amd_fractional_decimator_ff replays the CLI's loop (csdr.c:1511-1524: the function is called on the_bufsize-sample windows,
 * the unprocessed tail is re-presented) over the given input instead of one call over the whole array; 0 (default) = one call = the library function */
void  csdr_amd_fracdec_set_cli_bufsize(csdr_amd_fracdec *d, int the_bufsize);
float csdr_amd_fracdec_get_where(const csdr_amd_fracdec *d);

/* ------------------------------------------------------------------ f2: the remaining simple blocks
 * amdemod_cf / amdemod_estimator_cf libcsdr.c:861-901, realpart_cf csdr.c:634-645, logpower_cf libcsdr.c:1296-1303: flat arrays */
int csdr_amd_amdemod_cf(csdr_amd_ctx *ctx, const csdr_complexf *in, float *out, size_t n);
int csdr_amd_amdemod_estimator_cf(csdr_amd_ctx *ctx, const csdr_complexf *in, float *out, size_t n, float alpha, float beta);
int csdr_amd_realpart_cf(csdr_amd_ctx *ctx, const csdr_complexf *in, float *out, size_t n);
int csdr_amd_logpower_cf(csdr_amd_ctx *ctx, const csdr_complexf *in, float *out, size_t n, float add_db);
/* fmdemod_atan_cf libcsdr.c:1004-1019.  last_phase_io: device float[n_streams] */
int csdr_amd_fmdemod_atan_cf(csdr_amd_ctx *ctx, const csdr_complexf *in, float *out, int n_streams, size_t n,
                             size_t in_pitch, size_t out_pitch, float *last_phase_io);
/* dcblock_ff libcsdr.c:903-918.  state_io: device float[2*n_streams] = {last_input, last_output} (dcblock_preserve_t, libcsdr.h:110-114) */
int csdr_amd_dcblock_ff(csdr_amd_ctx *ctx, const float *in, float *out, int n_streams, size_t n, size_t in_pitch, size_t out_pitch,
                        float a, float *state_io);
/* fastdcblock_ff libcsdr.c:920-941 over n_blocks consecutive blocks per stream.  last_dc_io: device float[n_streams] */
int csdr_amd_fastdcblock_ff(csdr_amd_ctx *ctx, const float *in, float *out, int n_streams, int n_blocks, int block,
                            size_t in_pitch, size_t out_pitch, float *last_dc_io);
/* agc_ff libcsdr_gpl.c:163-260: n samples per stream processed as consecutive calls of `block` samples (the CLI's the_bufsize,
 * csdr.c:1338-1375: the hang/attack counters restart with every call).  last_gain_io: device float[n_streams] */
int csdr_amd_agc_ff(csdr_amd_ctx *ctx, const float *in, float *out, int n_streams, size_t n, int block, size_t in_pitch, size_t out_pitch,
                    float reference, float attack_rate, float decay_rate, float max_gain, short hang_time, short attack_wait_time,
                    float gain_filter_alpha, float *last_gain_io);
/* precalculate_window libcsdr.c:1256-1267 (host) */
void csdr_amd_precalculate_window(float *host_windowt, int size, int window);
/* `csdr fft_cc <fft_size> <out_of_every_n_samples> [window]` (csdr.c:1569-1641): windowed FFT of the newest fft_size samples every
 * every_n_samples new samples; the sliding buffer is carried inside the object.  Returns the number of spectra written. */
typedef struct csdr_amd_fftcc csdr_amd_fftcc;
csdr_amd_fftcc *csdr_amd_fftcc_create(csdr_amd_ctx *ctx, int fft_size, int every_n_samples, int window, int max_frames);
void csdr_amd_fftcc_destroy(csdr_amd_fftcc *f);
int  csdr_amd_fftcc_process(csdr_amd_fftcc *f, const csdr_complexf *in, size_t n_in, csdr_complexf *out, size_t *consumed);

/* ------------------------------------------------------------------ f3: IMA ADPCM (ima_adpcm.c:110-174), bit exact.
 * A serial state machine per stream: one lane per stream.  state_io: device int32[2*n_streams] = {index, previousValue} (ima_adpcm_state_t).
 * encode: n int16 samples per stream -> n/2 bytes (low nibble first; an odd last sample is dropped like the reference does);
 * decode: n bytes per stream -> 2n samples. */
int csdr_amd_encode_ima_adpcm_i16_u8(csdr_amd_ctx *ctx, const int16_t *in, uint8_t *out, int n_streams, size_t n,
                                     size_t in_pitch, size_t out_pitch, int *state_io);
int csdr_amd_decode_ima_adpcm_u8_i16(csdr_amd_ctx *ctx, const uint8_t *in, int16_t *out, int n_streams, size_t n,
                                     size_t in_pitch, size_t out_pitch, int *state_io);
/* `csdr compress_fft_adpcm_f_u8 <fft_size>` (csdr.c:1745-1768): n_blocks rows of fft_size dB values -> (fft_size+10)/2 bytes each; the
 * encoder restarts from the zero state for every row, so rows are independent (one lane per row). */
int csdr_amd_compress_fft_adpcm_f_u8(csdr_amd_ctx *ctx, const float *in, uint8_t *out, int n_blocks, int fft_size);

/* ------------------------------------------------------------------ FFT overlap-add filter
 * bandpass_fir_fft_cc (csdr.c:1810-1886) = apply_fir_fft_cc (libcsdr.c:814-849) per block.
 * One object per (fft_size, taps); processes n_blocks blocks of input_size = fft_size-taps_length+1 samples
 * for n_streams streams per call; the inter-block overlap is carried inside the object per stream. */
typedef struct csdr_amd_fftfilt csdr_amd_fftfilt;
csdr_amd_fftfilt *csdr_amd_fftfilt_create(csdr_amd_ctx *ctx, int fft_size, const csdr_complexf *host_taps,
                                          int taps_length, int n_streams, int max_blocks);
void csdr_amd_fftfilt_destroy(csdr_amd_fftfilt *f);
int  csdr_amd_fftfilt_set_taps(csdr_amd_fftfilt *f, const csdr_complexf *host_taps, int taps_length);
int  csdr_amd_fftfilt_input_size(const csdr_amd_fftfilt *f);
/* Taps short enough for windows that fit a CU's LDS (<= 4096 taps) are served by ONE pass over HBM (fftfilt_lds.hip: overlap-save with 4096 / 8192 / 16384-point
 * transforms in LDS; same samples out, same framing at this interface): its kernel name and window size, or "" / 0 when the full-size transform path runs. */
const char *csdr_amd_fftfilt_kernel_name(const csdr_amd_fftfilt *f);
int  csdr_amd_fftfilt_window(const csdr_amd_fftfilt *f);
int  csdr_amd_fftfilt_reset(csdr_amd_fftfilt *f);
int  csdr_amd_fftfilt_process(csdr_amd_fftfilt *f, const csdr_complexf *in, csdr_complexf *out,
                              int n_blocks, size_t in_pitch, size_t out_pitch);

/* building blocks used by the drop-in apply_fir_fft_cc / make_fft_c2c (libcsdr.c:814-849, fft_fftw.c:6-45) */
int csdr_amd_fft_c2c(csdr_amd_ctx *ctx, const csdr_complexf *in, csdr_complexf *out, int n, int forward);  /* unnormalised, cached plan */
int csdr_amd_bin_product(csdr_amd_ctx *ctx, const csdr_complexf *a, const csdr_complexf *b, csdr_complexf *out, size_t n);
/* io[k] = io[k]*scale (+ add[k] for k < n_add) */
int csdr_amd_scale_add(csdr_amd_ctx *ctx, csdr_complexf *io, size_t n, float scale, const csdr_complexf *add, size_t n_add);

/* ------------------------------------------------------------------ fastddc channelizer
 * fastddc_init fastddc.c:38-72 (layout == fastddc_t fastddc.h:5-24, 76 bytes) */
typedef struct csdr_fastddc_s {
    int pre_decimation, post_decimation, taps_length, taps_min_length, overlap_length,
        fft_size, fft_inv_size, input_size, post_input_size;
    float pre_shift; int startbin, v, offsetbin; float post_shift; int output_scrape, scrap;
    struct { float sindelta, cosdelta, rate; } dsadata;
} csdr_fastddc_t;
int csdr_amd_fastddc_init(csdr_fastddc_t *ddc, float transition_bw, int decimation, float shift_rate);

/* forward half = `csdr fastddc_fwd_cc` (csdr.c:2255-2300): overlap-save framing + fft_size forward FFT.
 * in: n_blocks*input_size new samples; spectra: [n_blocks][fft_size].  The overlap tail is kept in the object. */
typedef struct csdr_amd_fastddc_fwd csdr_amd_fastddc_fwd;
csdr_amd_fastddc_fwd *csdr_amd_fastddc_fwd_create(csdr_amd_ctx *ctx, const csdr_fastddc_t *ddc, int max_blocks);
void csdr_amd_fastddc_fwd_destroy(csdr_amd_fastddc_fwd *f);
int  csdr_amd_fastddc_fwd_process(csdr_amd_fastddc_fwd *f, const csdr_complexf *in, csdr_complexf *spectra, int n_blocks);

/* inverse half = `csdr fastddc_inv_cc` x n_channels (csdr.c:2302-2378, fastddc.c:106-166): per channel
 * alias-fold X*H into fft_inv_size bins, small inverse FFT, scrap, residual shift + post-decimation.
 * All channels share decimation/transition_bw (hence geometry) and differ by shift_rate.
 * out: [n_channels][out_pitch]; out_counts (host int[n_channels], may be NULL) receives samples written. */
typedef struct csdr_amd_fastddc_inv csdr_amd_fastddc_inv;
csdr_amd_fastddc_inv *csdr_amd_fastddc_inv_create(csdr_amd_ctx *ctx, float transition_bw, int decimation,
                                                  const float *host_shift_rates, int n_channels, int window,
                                                  int max_blocks);
void csdr_amd_fastddc_inv_destroy(csdr_amd_fastddc_inv *f);
/* retune one channel between two process() calls: geometry, taps and shift status of that channel are rebuilt (csdr.c:2329-2376) */
int  csdr_amd_fastddc_inv_set_rate(csdr_amd_fastddc_inv *f, int channel, float shift_rate);
int  csdr_amd_fastddc_inv_geometry(const csdr_amd_fastddc_inv *f, int channel, csdr_fastddc_t *ddc);
int  csdr_amd_fastddc_inv_max_output(const csdr_amd_fastddc_inv *f, int n_blocks);
int  csdr_amd_fastddc_inv_process(csdr_amd_fastddc_inv *f, const csdr_complexf *spectra, int n_blocks,
                                  csdr_complexf *out, size_t out_pitch, int *out_counts);
/* name of the dominant kernel ("k_ddc_gemm": the alias fold as an fp32 matrix-core product, taken at fft_inv_size 512 = BASELINE config 4;
 * "k_ddc_fold_ct": the general kernel) and HIP-event timing of it on the context's stream (bench_fastddc.py's roofline leg) */
const char *csdr_amd_fastddc_inv_kernel_name(const csdr_amd_fastddc_inv *f);
int  csdr_amd_fastddc_inv_set_profiling(csdr_amd_fastddc_inv *f, int on);
int  csdr_amd_fastddc_inv_kernel_time(csdr_amd_fastddc_inv *f, double *total_ms, long *launches);
/* the same for the kernels around it (csdr_amd_fastddc_inv_set_profiling(f, 2): two more event pairs per call): stage 1 = the forward transform's first pass (k_ddc_fwd512), stage 2 = the inverse transforms with scrap and residual
 * shift (k_ddc_ifft256d_post / k_ddc_ifft512_post) */
int  csdr_amd_fastddc_inv_stage_time(csdr_amd_fastddc_inv *f, int stage, double *total_ms, long *launches);
/* Both halves in one object = the ddcd topology (ddcd_old.cpp:238-252, 474-492: one `csdr fastddc_fwd_cc` feeding N `csdr fastddc_inv_cc --fd` clients)
 * as one call per batch of blocks: in = n_blocks * input_size NEW wideband samples (the overlap is kept inside), out / out_counts as for
 * csdr_amd_fastddc_inv_process.  At BASELINE config 4's geometry (fft_size 65536, fft_inv_size 512) the forward transform writes the fold's own layout
 * directly (no natural-order spectrum, no framing copy). */
typedef struct csdr_amd_fastddc_bank csdr_amd_fastddc_bank;
csdr_amd_fastddc_bank *csdr_amd_fastddc_bank_create(csdr_amd_ctx *ctx, float transition_bw, int decimation, const float *host_shift_rates, int n_channels,
                                                    int window, int max_blocks);
void csdr_amd_fastddc_bank_destroy(csdr_amd_fastddc_bank *b);
int  csdr_amd_fastddc_bank_set_rate(csdr_amd_fastddc_bank *b, int channel, float shift_rate);   /* csdr.c:2329-2376 for one client */
int  csdr_amd_fastddc_bank_input_size(const csdr_amd_fastddc_bank *b);
int  csdr_amd_fastddc_bank_max_output(const csdr_amd_fastddc_bank *b, int n_blocks);
int  csdr_amd_fastddc_bank_process(csdr_amd_fastddc_bank *b, const csdr_complexf *in, int n_blocks, csdr_complexf *out, size_t out_pitch, int *out_counts);
/* The same in two halves, so that consecutive batches overlap: submit() stages a batch (state chains, forward transform -- and, for a sharded bank, the
 * exchange -- on a side stream; `in` must stay untouched until the matching collect() has been queued); collect() folds / inverse-transforms the oldest staged
 * batch on the context's stream.  Two batches may be staged: submit(N + 1) runs under collect(N).  process() = submit() + collect(). */
int  csdr_amd_fastddc_bank_submit(csdr_amd_fastddc_bank *b, const csdr_complexf *in, int n_blocks);
int  csdr_amd_fastddc_bank_collect(csdr_amd_fastddc_bank *b, csdr_complexf *out, size_t out_pitch, int *out_counts);

/* ------------------------------------------------------------------ multi-GPU: one process per GPU, RCCL over xGMI
 * The communicator belongs to the library (RCCL is loaded on demand, librccl.so.1).  Rank 0 calls csdr_amd_comm_unique_id and ships the 128 bytes to
 * the other ranks by any means (torch.distributed / MPI / a file); every rank then calls csdr_amd_comm_create (collective, ncclCommInitRank). */
typedef struct csdr_amd_comm csdr_amd_comm;
int  csdr_amd_comm_unique_id(char id128[128]);
csdr_amd_comm *csdr_amd_comm_create(csdr_amd_ctx *ctx, const char id128[128], int rank, int world);
void csdr_amd_comm_destroy(csdr_amd_comm *c);
int  csdr_amd_comm_rank(const csdr_amd_comm *c);
int  csdr_amd_comm_world(const csdr_amd_comm *c);
int  csdr_amd_comm_broadcast(csdr_amd_comm *c, void *dev_buf, size_t bytes, int root);        /* on the context's stream (test / bench plumbing) */
/* A second communicator over the same ranks (collective; RCCL: a new unique id broadcast over the parent, then ncclCommInitRank).  The time-sliced bank makes one
 * for its output exchange, so that its two exchanges -- issued from two side streams -- never share an ncclComm.  Destroy with csdr_amd_comm_destroy. */
csdr_amd_comm *csdr_amd_comm_dup(csdr_amd_comm *c);
/* First contact with a transport / a box (collective): a send/recv ring, an all-gather, a broadcast and a two-communicator / two-stream ring of rank-stamped
 * buffers of n_floats floats each, every byte checked, timed.  report (may be NULL) receives one line per rank; CSDR_AMD_COMM_VERBOSE=1 also prints it on stderr.
 * 0 = every byte arrived as stamped; -6 otherwise (csdr_amd_last_error() holds the line). */
int  csdr_amd_comm_selftest(csdr_amd_comm *c, size_t n_floats, char *report, size_t report_cap);
/* Two more transports behind the same communicator type, for boxes with ONE GPU:
 * loopback -- the `world` ranks live in one process, one host thread + one context each (all on one device, or on several); every exchange is a
 *   stream-ordered device copy.  The multi-rank code of the channelizer runs unchanged (tests at world 2 / 4 / 8 on one MI355X).  Every rank thread
 *   must make the same exchange calls in the same order; a rank that never arrives fails the others after 60 s instead of hanging them.
 * null -- rank `rank` of a `world`-rank schedule with NO peers: every exchange call returns at once and moves nothing.  For timing one rank's own
 *   work of a world-N schedule on one GPU (bench_fastddc.py --emulate-world); the outputs are meaningless. */
typedef struct csdr_amd_loopback csdr_amd_loopback;
csdr_amd_loopback *csdr_amd_loopback_create(int world);
void csdr_amd_loopback_destroy(csdr_amd_loopback *g);
void csdr_amd_loopback_abort(csdr_amd_loopback *g);                 /* a rank thread gave up: fail the others' rendezvous at once */
csdr_amd_comm *csdr_amd_comm_create_loopback(csdr_amd_ctx *ctx, csdr_amd_loopback *g, int rank);
csdr_amd_comm *csdr_amd_comm_create_null(csdr_amd_ctx *ctx, int rank, int world);
/* The ranks as PROCESSES on one box (RCCL refuses two ranks per device; the loopback's ranks are threads): unix sockets named "<path_prefix>.<rank>" + HIP IPC mappings
 * of the peers' buffers.  A test transport (every group is host synchronous) for what only separate processes exercise: `csdr fastddc_bank_cc` started once per rank
 * (CSDR_AMD_COMM=ipc with CSDR_AMD_RANK / _WORLD / _COMM_FILE).  Collective; 60 s for the peers to appear. */
csdr_amd_comm *csdr_amd_comm_create_ipc(csdr_amd_ctx *ctx, const char *path_prefix, int rank, int world);
/* test hook: one send/receive group with `peer` (n floats each way; n_sends = 2 provokes the local error whose handling comm.cpp's l_group_end documents) */
int csdr_amd_debug_comm_exchange(csdr_amd_comm *c, const float *send_buf, float *recv_buf, size_t n, int peer, int n_sends);
/* fastddc bank over the communicator (BASELINE config 4 at 2 / 4 / 8 GPUs): host_shift_rates_all = ALL channels on every rank; rank r DELIVERS the
 * block-distributed slice of the channels csdr_amd_fastddc_bank_channel_slice reports (out rows = that slice; a channel's client connects to that GPU).
 * Two ways of dividing the work (shard_mode):
 *   CSDR_AMD_SHARD_BLOCKS (what csdr_amd_fastddc_bank_create_sharded picks for more than two ranks): time slices.  The blocks of a batch are dealt to the ranks in runs of
 *     ceil(max_blocks / world); every rank runs the whole single-GPU pipeline -- forward transform, fold of ALL channels, inverse transforms -- on its run and
 *     only the decimated outputs cross the links (all-to-all: each rank sends every peer that peer's channels of its run; 8 B per input sample in total, 1/world
 *     of it per link).  Possible because the only state that crosses block boundaries, decimating_shift_addition_cc's (remain, phase) per channel
 *     (fastddc.c:152-164), is data independent: every rank walks it over the whole batch itself.  The spectra never leave the GPU that computed them.
 *   CSDR_AMD_SHARD_CHANNELS (BASELINE north_star's partitioning; what csdr_amd_fastddc_bank_create_sharded picks for one or two ranks): the compute is channel-sharded
 *     as well: forward transform split by blocks, transposed spectra all-gathered over the full mesh (9.1 B per input sample arrive at EVERY rank), every rank folds
 *     its own channels.  Link bound beyond two GPUs.
 * csdr_amd_fastddc_bank_default_shard_mode(world) tells which one csdr_amd_fastddc_bank_create_sharded picks; csdr_amd_fastddc_bank_create_sharded_by takes either.
 * Input: submit / process read `in` on rank 0 only (the wideband stream lives there, like ddcd's single fastddc_fwd_cc, ddcd_old.cpp:238-252) and send every
 * rank the samples of its blocks point to point; csdr_amd_fastddc_bank_submit_local (time slices only) takes each rank's OWN run instead -- overlap_length
 * samples of the stream in front of the run's first block (zeros at the start of the stream), then its blocks -- when the ingest already distributes the
 * stream (csdr_amd_fastddc_bank_local_blocks tells a rank which blocks of an n-block batch are its run).
 * Needs the geometry of the matrix-core path (fft_size 65536, fft_inv_size 512).  All ranks make the same calls in the same order.
 * A time-sliced bank finishes a batch on its exchange stream: `out` is complete once csdr_amd_fastddc_bank_finish has been called (it orders the
 * context's stream behind the exchange; with out_counts it also waits and returns the sample counts).  collect() with out_counts != NULL and process()
 * call it themselves. */
enum { CSDR_AMD_SHARD_CHANNELS = 0, CSDR_AMD_SHARD_BLOCKS = 1 };
csdr_amd_fastddc_bank *csdr_amd_fastddc_bank_create_sharded(csdr_amd_ctx *ctx, float transition_bw, int decimation, const float *host_shift_rates_all, int n_channels_total,
                                                            int window, int max_blocks, csdr_amd_comm *comm);
csdr_amd_fastddc_bank *csdr_amd_fastddc_bank_create_sharded_by(csdr_amd_ctx *ctx, float transition_bw, int decimation, const float *host_shift_rates_all, int n_channels_total,
                                                               int window, int max_blocks, csdr_amd_comm *comm, int shard_mode);
int  csdr_amd_fastddc_bank_channel_slice(const csdr_amd_fastddc_bank *b, int *first, int *count);
int  csdr_amd_fastddc_bank_shard_mode(const csdr_amd_fastddc_bank *b);                          /* -1: one GPU */
int  csdr_amd_fastddc_bank_default_shard_mode(int world);
int  csdr_amd_fastddc_bank_local_blocks(const csdr_amd_fastddc_bank *b, int n_blocks, int *first, int *count);
int  csdr_amd_fastddc_bank_overlap(const csdr_amd_fastddc_bank *b);
int  csdr_amd_fastddc_bank_submit_local(csdr_amd_fastddc_bank *b, const csdr_complexf *in_run, int n_blocks);
/* Integer ingest: the wideband stream as s16 or u8 IQ pairs (what an SDR delivers; the reference puts convert_s16_f / convert_u8_f in front of fastddc_fwd_cc,
 * README.md:66-87, libcsdr.c:2363-2437, csdr.c:2255-2300).  The conversion runs inside the forward transform with the converters' own arithmetic -- the outputs
 * are bit-equal to csdr_amd_convert_s16_f / _u8_f followed by the complexf entry points -- and the stream crosses PCIe (and, in a sharded bank, the root's xGMI
 * links) at 4 or 2 bytes per sample instead of 8.  Matrix-core geometry only (fft 65536 / inverse 512: BASELINE config 4); other geometries: convert first. */
int  csdr_amd_fastddc_bank_process_s16(csdr_amd_fastddc_bank *b, const int16_t *in_iq, int n_blocks, csdr_complexf *out, size_t out_pitch, int *out_counts);
int  csdr_amd_fastddc_bank_process_u8(csdr_amd_fastddc_bank *b, const uint8_t *in_iq, int n_blocks, csdr_complexf *out, size_t out_pitch, int *out_counts);
int  csdr_amd_fastddc_bank_submit_s16(csdr_amd_fastddc_bank *b, const int16_t *in_iq, int n_blocks);
int  csdr_amd_fastddc_bank_submit_u8(csdr_amd_fastddc_bank *b, const uint8_t *in_iq, int n_blocks);
int  csdr_amd_fastddc_bank_submit_local_s16(csdr_amd_fastddc_bank *b, const int16_t *in_run_iq, int n_blocks);
int  csdr_amd_fastddc_bank_submit_local_u8(csdr_amd_fastddc_bank *b, const uint8_t *in_run_iq, int n_blocks);
int  csdr_amd_fastddc_bank_finish(csdr_amd_fastddc_bank *b, int *out_counts);
/* retune by GLOBAL channel number; every rank makes the same call (csdr_amd_fastddc_bank_set_rate takes an index into the rank's own slice; in a time-sliced
 * bank, where every rank computes every channel, a call on one rank only changes that rank's contribution).  A retune takes effect from the next SUBMITTED batch on (csdr.c:2329-2376: the new rate
 * between two reads): a time-sliced bank holds a retune that arrives while batches are staged back until they are collected; every other bank REFUSES it then
 * (-3: collect first) -- part of a staged batch's tables is already fixed. */
int  csdr_amd_fastddc_bank_set_rate_global(csdr_amd_fastddc_bank *b, int channel, float shift_rate);
/* the bank's inverse half (kernel name / profiling: csdr_amd_fastddc_inv_kernel_name, _set_profiling, _kernel_time) */
csdr_amd_fastddc_inv *csdr_amd_fastddc_bank_inverse(csdr_amd_fastddc_bank *b);

/* one channel, one block, explicit taps_fft and status = fastddc_inv_cc itself (fastddc.c:106-166).
 * status_io: HOST {decimation_remain, float starting_phase, output_size} (decimating_shift_addition_status_t).
 * d_inv_in / d_td: device scratch of fft_inv_size complexf (folded bins after the second swap / IFFT output, unnormalised). */
int  csdr_amd_fastddc_inv_block(csdr_amd_ctx *ctx, const csdr_complexf *d_spectrum, const csdr_complexf *d_taps_fft,
                                const csdr_fastddc_t *ddc, void *status_io, csdr_complexf *d_inv_in, csdr_complexf *d_td,
                                csdr_complexf *d_out);

/* ------------------------------------------------------------------ fused WFM receive chain (BASELINE config 2)
 * README.md:66 / csdr-fm:41:  convert_u8_f | shift_addition_cc r | fir_decimate_cc D tbw HAMMING |
 * fmdemod_quadri_cf | fractional_decimator_ff R | deemphasis_wfm_ff fs tau | convert_f_s16
 * for n_streams independent u8 IQ streams in one pass: each input byte is read from HBM once, each
 * audio sample written once.  Streaming: call repeatedly with consecutive blocks; all cross-block state
 * (shift phase, FIR/demod history, de-emphasis state, decimator position) lives in the object.
 * Scope: an FM AUDIO chain.  Its matrix-core front end models the reference's float rotator as C_m D^k per 1024-sample chunk; for rates at which
 * the reference's recurrence (libcsdr_gpl.c:44-45) drifts from that model (1e-5 .. 4e-5 per chunk at e.g. 0.05, 0.25) the deviation is a slowly
 * varying complex factor common to y[k] and y[k-1], which fmdemod_quadri_cf cancels: the audio is within the stated tolerance at every rate
 * (tests/test_edges_gpu.py::test_wfm_other_shift_rates), but the object's INTERNAL complex samples are not a substitute for
 * `convert_u8_f | shift_addition_cc | fir_decimate_cc` at such rates.  The complex front end for that is csdr_amd_ddc_* below, which replays the
 * drift per chunk (k_ddc_corr) and is held to 1e-5 on the complex samples themselves. */
typedef struct csdr_amd_wfm csdr_amd_wfm;
csdr_amd_wfm *csdr_amd_wfm_create(csdr_amd_ctx *ctx, int n_streams, float shift_rate, int decimation,
                                  const float *host_taps, int taps_length, int frac_rate,
                                  float tau, int audio_rate, size_t max_block_samples);
/* The same with a shift rate PER STREAM and its retune between calls (the reference's unit of work is one (stream, shift_rate) pair: `csdr shift_addition_cc --fifo`,
 * csdr.c:881-923; ddcd's per-client chains, ddcd_old.h:51-61) -- see csdr_amd_ddc_create_rates.  The chain kernel then gives a workgroup ONE stream and puts 16 time
 * segments of it into the 16 columns of the product; full rate needs blocks of >= 16 x lcm(4 D F, 1024) samples per stream and call (409 600 at D F = 50).
 * Needs the matrix-core kernel's shapes (csdr_amd_wfm_fallback == 0). */
csdr_amd_wfm *csdr_amd_wfm_create_rates(csdr_amd_ctx *ctx, int n_streams, const float *shift_rates, int decimation,
                                        const float *host_taps, int taps_length, int frac_rate,
                                        float tau, int audio_rate, size_t max_block_samples);
int   csdr_amd_wfm_set_rate(csdr_amd_wfm *w, int stream, float shift_rate);
float csdr_amd_wfm_get_rate(const csdr_amd_wfm *w, int stream);
void csdr_amd_wfm_destroy(csdr_amd_wfm *w);
int  csdr_amd_wfm_reset(csdr_amd_wfm *w);
/* in: u8 IQ, [n_streams][in_pitch bytes], block_samples complex samples per stream (multiple of 1024 except
 * for the last block of a stream).  audio_s16: [n_streams][out_pitch]; audio_f (optional, may be NULL)
 * receives the float audio before convert_f_s16 (parity tap).  Returns audio samples written per stream.
 * Any in_pitch that is a multiple of 16 bytes and any audio pointers / out_pitch are accepted; the streaming rate the benchmarks quote needs audio_f == NULL,
 * audio_s16 on a 16-byte boundary and out_pitch a multiple of 8 samples (the kernel then keeps the finished s16 lines in registers and stores 5.5 KiB per stream at
 * a time; other layouts are stored line by line or sample by sample: same values, 5-10 % slower). */
long csdr_amd_wfm_process(csdr_amd_wfm *w, const uint8_t *in, size_t in_pitch, size_t block_samples,
                          int16_t *audio_s16, float *audio_f, size_t out_pitch);
/* name and launch count of the dominant kernel of the last process() call (bench.py roofline leg) */
const char *csdr_amd_wfm_kernel_name(const csdr_amd_wfm *w);
/* 1 when the object runs outside the matrix-core chain kernel (k_wfm_front + k_wfm_back: decimation x audio decimation odd, filters beyond the 256-sample window;
 * CSDR_AMD_WFM_PATH=valu): same results at about a sixth of the rate.  The CLI prints a one-line note on stderr. */
int csdr_amd_wfm_fallback(const csdr_amd_wfm *w);
/* HIP-event timing of that kernel, on the context's stream: enable, run, then read the accumulated time. */
int csdr_amd_wfm_set_profiling(csdr_amd_wfm *w, int on);
int csdr_amd_wfm_kernel_time(csdr_amd_wfm *w, double *total_ms, long *launches);

/* ------------------------------------------------------------------ the RESIDENT form of the fused WFM chain: a persistent grid walking a ring of blocks
 * (north_star: "a persistent-kernel ring buffer so the stdin->stdout pipe never round-trips to host between stages").  The reference's unit of work is one
 * the_bufsize block -- 16384 samples -- per loop iteration of every stage (csdr.c:189-193, 232-247, 330-392); a kernel launch per such block spends most of its time
 * outside the data (launch gap, kernel entry, the de-emphasis warm-up of the time segments a short call is cut into).  A ring object owns
 *     an input ring   [n_slots][n_streams][pitch]  u8 IQ in device memory  -- block seq of every stream lies in slot seq mod n_slots,
 *     an output ring  [n_slots][n_streams][pitch]  s16 audio in device memory,
 * and ONE grid that stays on the GPU, polls block descriptors in host memory (no launch, no stream operation per block) and tells the host through a done word
 * per slot.  Same arithmetic and the same samples as csdr_amd_wfm_process on the same stream cut into the same blocks (tests/test_ring_gpu.py: >= 1000 consecutive
 * 16384-sample blocks with a retune in the middle against the oracle).  Protocol per block:
 *     seq = csdr_amd_wfm_ring_submitted(r);
 *     csdr_amd_wfm_ring_acquire(r, seq, 0);                      -- waits until slot seq mod n_slots may be overwritten (blocks seq - n_slots and seq - n_slots + 1 finished:
 *                                                                   a block's input stays the NEXT block's history)
 *     write [n_streams][2 block_samples] bytes to csdr_amd_wfm_ring_input(r, seq, &pitch)   (any producer: hipMemcpy, a kernel, a peer through HIP IPC) and complete it;
 *     csdr_amd_wfm_ring_submit(r);                               -- posts the block (a store to host memory; relaunches the grid if it has left)
 *     n = csdr_amd_wfm_ring_wait(r, seq, 0);                     -- audio samples per stream of that block, in csdr_amd_wfm_ring_output(r, seq, &pitch); valid until
 *                                                                   block seq + n_slots is posted.  Up to n_slots - 2 blocks may be in flight.
 * The grid never holds the GPU against a silent host: it leaves by itself when no block arrived for idle_us (default 200) or when it is older than life_ms (default
 * 250), and is relaunched by the next submit / wait; no state lives in it (every block warms its de-emphasis up over the 48 audio samples in front of it).  While it is
 * resident it occupies every CU it runs on (one workgroup per CU): other kernels of the process wait for it to leave.
 * block_samples: a multiple of 1024, 4096 .. 65536.  Retune: csdr.c:881-923 semantics (from the next posted block's first sample, phase carried). */
typedef struct csdr_amd_wfm_ring csdr_amd_wfm_ring;
csdr_amd_wfm_ring *csdr_amd_wfm_ring_create(csdr_amd_ctx *ctx, int n_streams, float shift_rate, int decimation, const float *host_taps, int taps_length, int frac_rate,
                                            float tau, int audio_rate, size_t block_samples, int n_slots);
void csdr_amd_wfm_ring_destroy(csdr_amd_wfm_ring *r);
int  csdr_amd_wfm_ring_reset(csdr_amd_wfm_ring *r);                                    /* stream start: block 0 next, phase 0 */
int  csdr_amd_wfm_ring_acquire(csdr_amd_wfm_ring *r, long long seq, double timeout_s);  /* timeout_s <= 0: 10 s */
uint8_t *csdr_amd_wfm_ring_input(csdr_amd_wfm_ring *r, long long seq, size_t *pitch_bytes);
long long csdr_amd_wfm_ring_submit(csdr_amd_wfm_ring *r);                              /* returns the block's sequence number, < 0 on error */
long csdr_amd_wfm_ring_wait(csdr_amd_wfm_ring *r, long long seq, double timeout_s);    /* audio samples per stream, < 0 on error / timeout */
const int16_t *csdr_amd_wfm_ring_output(csdr_amd_wfm_ring *r, long long seq, size_t *pitch_samples);
int  csdr_amd_wfm_ring_set_rate(csdr_amd_wfm_ring *r, float shift_rate);               /* drains the ring, rebuilds the weight set; from the next posted block on */
float csdr_amd_wfm_ring_get_rate(const csdr_amd_wfm_ring *r);
int  csdr_amd_wfm_ring_set_timeouts(csdr_amd_wfm_ring *r, double idle_us, double life_ms);
int  csdr_amd_wfm_ring_resident(csdr_amd_wfm_ring *r);                                 /* 1 while the grid is on the GPU */
int  csdr_amd_wfm_ring_stop(csdr_amd_wfm_ring *r);                                     /* asks the grid to leave and waits for it (posted blocks are finished first) */
int  csdr_amd_wfm_ring_slots(const csdr_amd_wfm_ring *r);
int  csdr_amd_wfm_ring_grid(const csdr_amd_wfm_ring *r);                               /* workgroups of the resident grid */
long csdr_amd_wfm_ring_launches(const csdr_amd_wfm_ring *r);                           /* launches of the grid so far */
long long csdr_amd_wfm_ring_submitted(const csdr_amd_wfm_ring *r);
/* where the grid's time went since the last reset (stops the grid): out[0..2] = microseconds per work item (block x 16-stream group) spent waiting for a block, in the
 * chain's body, in the completion; out[3] = items */
int  csdr_amd_wfm_ring_stats(csdr_amd_wfm_ring *r, double out[4]);
/* benchmark / soak aid: posts n_blocks blocks whose inputs are what lies in the ring's slots, as fast as the ring takes them, waits for the last one; t_first_us /
 * t_done_us (NULL or n_blocks doubles) receive every block's start / completion on the device clock */
int  csdr_amd_wfm_ring_replay(csdr_amd_wfm_ring *r, long n_blocks, double *t_first_us, double *t_done_us);
/* device clock (microseconds, arbitrary origin) at which the first workgroup took an item of block seq and at which its last item was finished */
int  csdr_amd_wfm_ring_block_times(csdr_amd_wfm_ring *r, long long seq, double *t_first_us, double *t_done_us);

/* ------------------------------------------------------------------ fused receiver front end (head of the NFM / AM / SSB chains, BASELINE config 5)
 * README.md:87, 95, 110:  convert_u8_f | shift_addition_cc r | fir_decimate_cc D tbw window
 * (libcsdr.c:2363-2368, libcsdr_gpl.c:27-52 in the CLI's 1024-sample chunks csdr.c:911-918, libcsdr.c:528-549 with the CLI's re-feed loop
 * csdr.c:1160-1176) for n_streams independent u8 IQ streams in one pass:  y[k] = sum_t taps[t] x'[D k + t].  Each input byte is read from HBM
 * once; the decimated complex stream is written once.  Streaming: consecutive blocks, all cross-block state (shift phase, FIR history,
 * output index) lives in the object.  taps_length <= 1025. */
typedef struct csdr_amd_ddc csdr_amd_ddc;
csdr_amd_ddc *csdr_amd_ddc_create(csdr_amd_ctx *ctx, int n_streams, float shift_rate, int decimation,
                                  const float *host_taps, int taps_length, size_t max_block_samples);
/* The same with a shift rate PER STREAM -- the reference's unit of work is one (stream, shift_rate) pair: `csdr shift_addition_cc --fifo` retunes a running stream
 * (csdr.c:881-923), ddcd starts one `csdr shift_unroll_cc --fd N | csdr fir_decimate_cc D bw` per client (ddcd_old.h:51-61).  shift_rates: n_streams floats.
 * The matrix-core kernel then gives a workgroup ONE stream and puts 16 time segments of it into the 16 columns of the product (the weights are the stream's own);
 * full rate needs blocks of >= 16 x lcm(8 D, 1024) samples per stream and call (409 600 at D = 50), shorter blocks fill fewer columns.
 * csdr_amd_ddc_set_rate: effective from the next call's first sample, the float phase carries over exactly as the reference's starting_phase does; the outputs of
 * that call whose window still reaches into samples rotated at the old rate are evaluated with both rates.  Returns 0 or a negative error. */
csdr_amd_ddc *csdr_amd_ddc_create_rates(csdr_amd_ctx *ctx, int n_streams, const float *shift_rates, int decimation,
                                        const float *host_taps, int taps_length, size_t max_block_samples);
int   csdr_amd_ddc_set_rate(csdr_amd_ddc *d, int stream, float shift_rate);
float csdr_amd_ddc_get_rate(const csdr_amd_ddc *d, int stream);
/* 1 when the last process() call ran outside the matrix-core kernel (k_ddc_direct: odd decimation, windows beyond 2304 bytes, input pitch not a multiple of 128 bytes,
 * ragged or very short blocks): same results, a fraction of the rate */
int   csdr_amd_ddc_fallback(const csdr_amd_ddc *d);
void csdr_amd_ddc_destroy(csdr_amd_ddc *d);
int  csdr_amd_ddc_reset(csdr_amd_ddc *d);
/* in: u8 IQ, [n_streams][in_pitch bytes], block_samples complex samples per stream (multiple of 1024 except for the last block of a
 * stream).  out: [n_streams][out_pitch] complexf.  Returns the number of outputs written per stream (all k with D k + taps <= samples so far). */
long csdr_amd_ddc_process(csdr_amd_ddc *d, const uint8_t *in, size_t in_pitch, size_t block_samples,
                          csdr_complexf *out, size_t out_pitch);
const char *csdr_amd_ddc_kernel_name(const csdr_amd_ddc *d);
int csdr_amd_ddc_set_profiling(csdr_amd_ddc *d, int on);
int csdr_amd_ddc_kernel_time(csdr_amd_ddc *d, double *total_ms, long *launches);
/* Test hook: CPU evaluation of one tile of the front end's matrix-core kernel with its own weight table, K-range split and chunk-boundary
 * handling (no GPU needed).  window: 2304 raw bytes from sample n0 (multiple of 16); ctab3: (cos, sin) of chunks n0>>10, +1, +2; out16: 8 x (Re, Im). */
int csdr_amd_debug_ddc_mfma_tile(int D, int L, float shift_rate, const float *taps, long long n0, const uint8_t *window,
                                 const float *ctab3, float *out16);

/* ------------------------------------------------------------------ NFM receive chain (BASELINE config 5)
 * README.md:87:  convert_u8_f | shift_addition_cc r | fir_decimate_cc D tbw HAMMING | fmdemod_quadri_cf | limit_ff [max] |
 *                deemphasis_nfm_ff fs | fastagc_ff [block [reference]] | convert_f_s16
 * for n_streams independent u8 IQ streams: front end = csdr_amd_ddc (one pass over the input), back end at the audio rate.  Streaming as the
 * CLI pipeline does it: the de-emphasis FIR re-feeds its unconsumed input (csdr.c:1083), fastagc_ff works on whole blocks with its two-block
 * latency and zero-initialised state (csdr.c:1393-1394); a call returns the whole AGC blocks that became available (a multiple of agc_block,
 * possibly 0).  audio_rate selects the de-emphasis table (48000, 44100, 8000, 11025; libcsdr.c:1115-1119). */
typedef struct csdr_amd_nfm csdr_amd_nfm;
csdr_amd_nfm *csdr_amd_nfm_create(csdr_amd_ctx *ctx, int n_streams, float shift_rate, int decimation, const float *host_taps, int taps_length,
                                  int audio_rate, int agc_block, float agc_reference, float limit_max, size_t max_block_samples);
/* a shift rate per channel, and its retune (csdr_amd_ddc_create_rates / csdr_amd_ddc_set_rate on the chain's front end) */
csdr_amd_nfm *csdr_amd_nfm_create_rates(csdr_amd_ctx *ctx, int n_streams, const float *shift_rates, int decimation, const float *host_taps, int taps_length,
                                        int audio_rate, int agc_block, float agc_reference, float limit_max, size_t max_block_samples);
int  csdr_amd_nfm_set_rate(csdr_amd_nfm *w, int stream, float shift_rate);
void csdr_amd_nfm_destroy(csdr_amd_nfm *w);
int  csdr_amd_nfm_reset(csdr_amd_nfm *w);
/* in: u8 IQ as for csdr_amd_ddc_process.  audio_s16: [n_streams][out_pitch]; audio_f (optional, may be NULL): the float audio before
 * convert_f_s16 (parity tap).  Returns audio samples written per stream. */
long csdr_amd_nfm_process(csdr_amd_nfm *w, const uint8_t *in, size_t in_pitch, size_t block_samples,
                          int16_t *audio_s16, float *audio_f, size_t out_pitch);
/* the chain's front end object (kernel name / profiling: csdr_amd_ddc_kernel_name, csdr_amd_ddc_set_profiling, csdr_amd_ddc_kernel_time) */
csdr_amd_ddc *csdr_amd_nfm_front_end(csdr_amd_nfm *w);

/* Test hook: one tile of the WFM chain kernel (k_wfm_mfma_seq: phase-independent weight set, post factors, chunk-boundary handling) on the CPU.
 * n0: window base sample (multiple of 8); window: 512 raw bytes; ctab2: (cos, sin) of chunks n0>>10, +1; out16: 16 rows. */
int csdr_amd_debug_wfm_seq_tile(int D, int L, int F, float shift_rate, const float *taps, long long n0, const uint8_t *window,
                                const float *ctab2, float *out16);
/* Test hook: the register-level 16-point butterfly of the three-pass 65536-point transform (fft64k.hip) on the CPU; 16 interleaved complex floats */
void csdr_amd_debug_dft16(const float *in32, float *out32, int inverse);
/* Test hook: the 8-point butterfly of the channelizer's 512-point inverse transforms (fastddc_mfma.hip) on the CPU; 8 interleaved complex floats */
int  csdr_amd_debug_fftfilt_lds(int n, const float *taps_iq, int taps_len, const float *x_iq, long m_new, float *y_iq);   /* CPU run of the one-pass filter kernel's stages */
void csdr_amd_debug_dft8(const float *in16, float *out16, int inverse);
/* Test hook: the channelizer's residual-shift bookkeeping (decimating_shift_addition_cc's (remain, phase) per block, libcsdr_gpl.c:153-158) over n_blocks blocks on the
 * CPU; mode 0 = the general step, mode 1 = the constant-step fast path of the kernels (-1 when it does not apply).  phases_out[b] = phase in front of block b. */
int  csdr_amd_debug_ddc_chain(int mode, float rate2, int post_in, int post_dec, int n_blocks, int *remain_io, float *phase_io, float *phases_out, int *count_out);
/* Test hook: one tile (16 outputs from 256 limited samples) of the NFM chain's matrix-core de-emphasis FIR on the CPU: digit planes,
 * Toeplitz digit table and accumulator classes as k_nfm_deemph_mfma combines them. */
int csdr_amd_debug_nfm_deemph_tile(int audio_rate, float max_amp, const float *x, float *out16);
#ifdef __cplusplus
}
#endif
#endif
