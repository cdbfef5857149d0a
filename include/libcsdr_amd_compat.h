/* include/libcsdr_amd_compat.h -- the reference's own C API, served by libcsdr_amd.so.
 *
 * Same symbol names, argument order, by-value state conventions and struct layouts as
 *   libcsdr.h:46-229, libcsdr_gpl.h:26-46, fastddc.h:5-29, fft_fftw.h:14-27
 * (layouts checked against the compiled reference in tests/test_abi.py), so a program
 * written against the reference headers links against libcsdr_amd.so unchanged.
 *
 * All pointers are HOST pointers: each call copies its block to the MI355X, runs the same HIP kernels as
 * the device batch API (include/csdr_amd.h) and copies the result back before returning, i.e. it appears
 * synchronous like the reference.  Per-call PCIe + launch latency makes this the compatibility path, not the
 * fast path; throughput work goes through the batch API / the csdr CLI shim.
 * Threads: like the reference's functions these keep no hidden state and may be called from several host threads at once; the library keeps one
 * device context (stream + staging) per calling thread, released when that thread exits.
 * There is NO CPU fallback: without a gfx950 device the first call prints the reason and aborts.
 */
#ifndef LIBCSDR_AMD_COMPAT_H
#define LIBCSDR_AMD_COMPAT_H
#include <stdio.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct complexf_s { float i; float q; } complexf;                          /* libcsdr.h:46 */
typedef enum window_s { WINDOW_BOXCAR, WINDOW_BLACKMAN, WINDOW_HAMMING } window_t;   /* libcsdr.h:70-73 */
#define WINDOW_DEFAULT WINDOW_HAMMING

/* FFT plan layer, fft_fftw.h:14-27 */
#define FFT_PLAN_T struct fft_plan_s
struct fft_plan_s { int size; void *input; void *output; void *plan; };
FFT_PLAN_T *make_fft_c2c(int size, complexf *input, complexf *output, int forward, int benchmark);
FFT_PLAN_T *make_fft_r2c(int size, float *input, complexf *output, int benchmark);
FFT_PLAN_T *make_fft_c2r(int size, complexf *input, float *output, int benchmark);
void fft_execute(FFT_PLAN_T *plan);
void fft_destroy(FFT_PLAN_T *plan);
void *csdr_fft_malloc(size_t n);      /* the reference's fft_malloc/fft_free are FFTW macros (fft_fftw.h:11-12) */
void csdr_fft_free(void *p);
#define fft_malloc csdr_fft_malloc
#define fft_free csdr_fft_free
void *fftwf_malloc(size_t n);          /* what the REFERENCE's fft_fftw.h:11-12 expands fft_malloc / fft_free to: exported too, so a client built against */
void fftwf_free(void *p);              /* the reference headers links without -lfftw3f */

/* filter design, libcsdr.h:85-92 */
void firdes_lowpass_f(float *output, int length, float cutoff_rate, window_t window);
void firdes_bandpass_c(complexf *output, int length, float lowcut, float highcut, window_t window);
float firdes_wkernel_blackman(float input);
float firdes_wkernel_hamming(float input);
float firdes_wkernel_boxcar(float input);
window_t firdes_get_window_from_string(char *input);
char *firdes_get_string_from_window(window_t window);
int firdes_filter_len(float transition_bw);
void normalize_fir_f(float *input, float *output, int length);

/* demodulators, libcsdr.h:95-100 */
complexf fmdemod_quadri_cf(complexf *input, float *output, int input_size, float *temp, complexf last_sample);
complexf fmdemod_quadri_novect_cf(complexf *input, float *output, int input_size, complexf last_sample);
void limit_ff(float *input, float *output, int input_size, float max_amplitude);

/* filters, decimators, shift, libcsdr.h:103-108 */
float fir_one_pass_ff(float *input, float *taps, int taps_length);
int fir_decimate_cc(complexf *input, complexf *output, int input_size, int decimation, float *taps, int taps_length);
int deemphasis_nfm_ff(float *input, float *output, int input_size, int sample_rate);
float deemphasis_wfm_ff(float *input, float *output, int input_size, float tau, int sample_rate, float last_output);
float shift_math_cc(complexf *input, complexf *output, int input_size, float rate, float starting_phase);

typedef struct fastagc_ff_s {                                                       /* libcsdr.h:118-128 */
    float *buffer_1; float *buffer_2; float *buffer_input;
    float peak_1; float peak_2; int input_size; float reference; float last_gain;
} fastagc_ff_t;
void fastagc_ff(fastagc_ff_t *input, float *output);

typedef struct fractional_decimator_ff_s {                                          /* libcsdr.h:151-168 */
    float where; int input_processed; int output_size; int num_poly_points;
    float *poly_precalc_denomiator; float *coeffs_buf; float *filtered_buf;
    int xifirst; int xilast; float rate; float *taps; int taps_length;
} fractional_decimator_ff_t;
fractional_decimator_ff_t fractional_decimator_ff_init(float rate, int num_poly_points, float *taps, int taps_length);
void fractional_decimator_ff(float *input, float *output, int input_size, fractional_decimator_ff_t *d);

typedef struct shift_table_data_s { float *table; int table_size; } shift_table_data_t;     /* libcsdr.h:180-184 */
void shift_table_deinit(shift_table_data_t table_data);
shift_table_data_t shift_table_init(int table_size);
float shift_table_cc(complexf *input, complexf *output, int input_size, float rate, shift_table_data_t table_data, float starting_phase);

typedef struct shift_addfast_data_s { float dsin[4]; float dcos[4]; float phase_increment; } shift_addfast_data_t;  /* :189-194 */
shift_addfast_data_t shift_addfast_init(float rate);
float shift_addfast_cc(complexf *input, complexf *output, int input_size, shift_addfast_data_t *d, float starting_phase);

typedef struct shift_unroll_data_s { float *dsin; float *dcos; float phase_increment; int size; } shift_unroll_data_t; /* :199-205 */
float shift_unroll_cc(complexf *input, complexf *output, int input_size, shift_unroll_data_t *d, float starting_phase);
shift_unroll_data_t shift_unroll_init(float rate, int size);

int log2n(int x);
int next_pow2(int x);
void apply_fir_fft_cc(FFT_PLAN_T *plan, FFT_PLAN_T *plan_inverse, complexf *taps_fft, complexf *last_overlap, int overlap_size);
void gain_ff(float *input, float *output, int input_size, float gain);

/* f2 blocks: libcsdr.h:97-99, 110-116, 142-147; libcsdr_gpl.h:37 */
float fmdemod_atan_cf(complexf *input, float *output, int input_size, float last_phase);
void amdemod_cf(complexf *input, float *output, int input_size);
void amdemod_estimator_cf(complexf *input, float *output, int input_size, float alpha, float beta);
typedef struct dcblock_preserve_s { float last_input; float last_output; } dcblock_preserve_t;             /* libcsdr.h:110-114 */
dcblock_preserve_t dcblock_ff(float *input, float *output, int input_size, float a, dcblock_preserve_t preserved);
float fastdcblock_ff(float *input, float *output, int input_size, float last_dc_level);
float *precalculate_window(int size, window_t window);
void apply_window_c(complexf *input, complexf *output, int size, window_t window);
void apply_precalculated_window_c(complexf *input, complexf *output, int size, float *windowt);
void logpower_cf(complexf *input, float *output, int size, float add_db);
float agc_ff(float *input, float *output, int input_size, float reference, float attack_rate, float decay_rate, float max_gain,
             short hang_time, short attack_wait_time, float gain_filter_alpha, float last_gain);

/* IMA ADPCM, ima_adpcm.h:5-11 */
typedef struct ImaState { int index; int previousValue; } ima_adpcm_state_t;
ima_adpcm_state_t encode_ima_adpcm_i16_u8(short *input, unsigned char *output, int input_length, ima_adpcm_state_t state);
ima_adpcm_state_t decode_ima_adpcm_u8_i16(unsigned char *input, short *output, int input_length, ima_adpcm_state_t state);

/* converters, libcsdr.h:220-229 */
void convert_u8_f(unsigned char *input, float *output, int input_size);
void convert_f_u8(float *input, unsigned char *output, int input_size);
void convert_s8_f(signed char *input, float *output, int input_size);
void convert_f_s8(float *input, signed char *output, int input_size);
void convert_f_s16(float *input, short *output, int input_size);
void convert_s16_f(short *input, float *output, int input_size);
void convert_f_i16(float *input, short *output, int input_size);
void convert_i16_f(short *input, float *output, int input_size);
void convert_f_s24(float *input, unsigned char *output, int input_size, int bigendian);
void convert_s24_f(unsigned char *input, float *output, int input_size, int bigendian);

/* libcsdr_gpl.h:26-46 */
typedef struct shift_addition_data_s { float sindelta; float cosdelta; float rate; } shift_addition_data_t;
shift_addition_data_t shift_addition_init(float rate);
float shift_addition_cc(complexf *input, complexf *output, int input_size, shift_addition_data_t d, float starting_phase);
float shift_addition_fc(float *input, complexf *output, int input_size, shift_addition_data_t d, float starting_phase);
typedef struct decimating_shift_addition_status_s { int decimation_remain; float starting_phase; int output_size; } decimating_shift_addition_status_t;
decimating_shift_addition_status_t decimating_shift_addition_cc(complexf *input, complexf *output, int input_size, shift_addition_data_t d, int decimation, decimating_shift_addition_status_t s);
shift_addition_data_t decimating_shift_addition_init(float rate, int decimation);

/* fastddc.h:5-29 */
typedef struct fastddc_s {
    int pre_decimation; int post_decimation; int taps_length; int taps_min_length; int overlap_length;
    int fft_size; int fft_inv_size; int input_size; int post_input_size;
    float pre_shift; int startbin; int v; int offsetbin; float post_shift; int output_scrape; int scrap;
    shift_addition_data_t dsadata;
} fastddc_t;
int fastddc_init(fastddc_t *ddc, float transition_bw, int decimation, float shift_rate);
decimating_shift_addition_status_t fastddc_inv_cc(complexf *input, complexf *output, fastddc_t *ddc, FFT_PLAN_T *plan_inverse, complexf *taps_fft, decimating_shift_addition_status_t shift_stat);
void fastddc_print(fastddc_t *ddc, char *source);
void fft_swap_sides(complexf *io, int fft_size);

#ifdef __cplusplus
}
#endif
#endif
