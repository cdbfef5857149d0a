/* oracle/csdr_oracle.h -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * CPU restatement (plain C, IEEE float, no fast-math, no FMA contraction) of the
 * libcsdr block-streaming DSP hot path, written from the behaviour of the reference
 * at /root/reference (every function cites the file:line it follows).  It is the
 * checker for the HIP path: only tests/, __graft_entry__.smoke() and bench.py's
 * cpu_baseline leg may load it.  The product library (csdr_amd/csrc) never does.
 *
 * Parity status: PINNED.  The reference holds no golden vectors for this path
 * (SURVEY.md section 4), so the restatement is pinned against the reference itself,
 * compiled unmodified into oracle/_ref/libcsdr_ref.so (oracle/Makefile) and compared
 * function by function in tests/test_oracle_vs_ref.py; fixtures generated from that
 * build are committed under tests/golden/ (script: tests/golden/make_golden.py).
 */
#ifndef CSDR_ORACLE_H
#define CSDR_ORACLE_H
#ifdef __cplusplus
extern "C" {
#endif

typedef struct { float i, q; } orc_cf;                 /* libcsdr.h:46 complexf */
enum { ORC_BOXCAR = 0, ORC_BLACKMAN = 1, ORC_HAMMING = 2 };   /* libcsdr.h:70-73 */

/* ---- design helpers ---- */
int   orc_firdes_filter_len(float transition_bw);                                   /* libcsdr.c:169-174 */
void  orc_firdes_lowpass_f(float *taps, int length, float cutoff_rate, int window); /* libcsdr.c:117-142 */
void  orc_firdes_bandpass_c(orc_cf *taps, int length, float lowcut, float highcut, int window); /* :144-167 */
int   orc_next_pow2(int x);                                                         /* libcsdr.c:1235-1243 */
int   orc_log2n(int x);                                                             /* libcsdr.c:1220-1233 */

/* ---- sample format converters (bit-exact contract) ---- */
void orc_convert_u8_f(const unsigned char *in, float *out, int n);   /* libcsdr.c:2363-2366 */
void orc_convert_s8_f(const signed char *in, float *out, int n);     /* :2368-2371 */
void orc_convert_s16_f(const short *in, float *out, int n);          /* :2373-2376 */
void orc_convert_f_u8(const float *in, unsigned char *out, int n);   /* :2378-2383 */
void orc_convert_f_s8(const float *in, signed char *out, int n);     /* :2385-2388 */
void orc_convert_f_s16(const float *in, short *out, int n);          /* :2390-2398 */
void orc_convert_f_s24(const float *in, unsigned char *out, int n, int bigendian); /* :2403-2423 */
void orc_convert_s24_f(const unsigned char *in, float *out, int n, int bigendian); /* :2425-2437 */

/* ---- frequency shifters ---- */
typedef struct { float sindelta, cosdelta, rate; } orc_shift_addition_t;           /* libcsdr_gpl.h:26-31 */
typedef struct { int decimation_remain; float starting_phase; int output_size; } orc_dsa_status_t; /* :39-44 */
float orc_shift_math_cc(const orc_cf *in, orc_cf *out, int n, float rate, float starting_phase); /* libcsdr.c:186-207 */
void  orc_shift_table_init(float *table, int table_size);                          /* libcsdr.c:211-222 */
float orc_shift_table_cc(const orc_cf *in, orc_cf *out, int n, float rate, const float *table, int table_size, float starting_phase); /* :229-265 */
float orc_shift_unroll_init(float rate, int size, float *dsin, float *dcos);       /* :268-284, returns phase_increment */
float orc_shift_unroll_cc(const orc_cf *in, orc_cf *out, int n, const float *dsin, const float *dcos, float phase_increment, float starting_phase); /* :286-305 */
float orc_shift_addfast_init(float rate, float *dsin4, float *dcos4);              /* :307-317, returns phase_increment */
float orc_shift_addfast_cc(const orc_cf *in, orc_cf *out, int n, const float *dsin4, const float *dcos4, float phase_increment, float starting_phase); /* :406-434 */
orc_shift_addition_t orc_shift_addition_init(float rate);                          /* libcsdr_gpl.c:81-89 */
float orc_shift_addition_cc(const orc_cf *in, orc_cf *out, int n, orc_shift_addition_t d, float starting_phase); /* :27-52 */
float orc_shift_addition_fc(const float *in, orc_cf *out, int n, orc_shift_addition_t d, float starting_phase);  /* :54-79 */
orc_shift_addition_t orc_decimating_shift_addition_init(float rate, int decimation); /* :126-129 */
orc_dsa_status_t orc_decimating_shift_addition_cc(const orc_cf *in, orc_cf *out, int n, orc_shift_addition_t d, int decimation, orc_dsa_status_t s); /* :131-160 */

/* ---- filters / demod / audio ---- */
int    orc_fir_decimate_cc(const orc_cf *in, orc_cf *out, int n, int decimation, const float *taps, int taps_length); /* libcsdr.c:528-549 */
orc_cf orc_fmdemod_quadri_cf(const orc_cf *in, float *out, int n, orc_cf last_sample);  /* :1021,1040-1071 */
float  orc_deemphasis_wfm_ff(const float *in, float *out, int n, float tau, int sample_rate, float last_output); /* :1081-1097 */
int    orc_deemphasis_nfm_ff(const float *in, float *out, int n, const float *taps, int taps_length); /* :1101-1128 (table passed in) */
void   orc_limit_ff(const float *in, float *out, int n, float max_amplitude);           /* :1130-1137 */
void   orc_gain_ff(const float *in, float *out, int n, float gain);                     /* :1139-1142 */

typedef struct {
    float *buffer_1, *buffer_2, *buffer_input;
    float peak_1, peak_2; int input_size; float reference, last_gain;
} orc_fastagc_t;                                                                        /* libcsdr.h:118-128 */
void orc_fastagc_ff(orc_fastagc_t *st, float *out);                                     /* libcsdr.c:946-991 */

typedef struct {
    float where; int input_processed, output_size, num_poly_points;
    float denom[64]; int xifirst, xilast; float rate; const float *taps; int taps_length;
} orc_fracdec_t;                                                                        /* libcsdr.h:151-168 */
void orc_fractional_decimator_ff_init(orc_fracdec_t *d, float rate, int num_poly_points, const float *taps, int taps_length); /* :715-748 */
void orc_fractional_decimator_ff(const float *in, float *out, int n, orc_fracdec_t *d); /* :751-793 */

/* ---- FFT overlap-add filter (one block) ---- */
/* in: fft_size complexf (input_size samples + zero pad); taps_fft: fft_size; last_overlap: overlap complexf;
 * result: fft_size complexf.  Follows libcsdr.c:814-849. */
void orc_apply_fir_fft_cc(const orc_cf *in, orc_cf *result, int fft_size, const orc_cf *taps_fft, const orc_cf *last_overlap, int overlap);
void orc_fft_c2c(const orc_cf *in, orc_cf *out, int n, int forward);                    /* fft_fftw.c:6-15,36-39 */

/* ---- fastddc ---- */
typedef struct {
    int pre_decimation, post_decimation, taps_length, taps_min_length, overlap_length,
        fft_size, fft_inv_size, input_size, post_input_size;
    float pre_shift; int startbin, v, offsetbin; float post_shift; int output_scrape, scrap;
    orc_shift_addition_t dsadata;
} orc_fastddc_t;                                                                        /* fastddc.h:5-24 */
int  orc_fastddc_init(orc_fastddc_t *ddc, float transition_bw, int decimation, float shift_rate); /* fastddc.c:38-72 */
void orc_fft_swap_sides(orc_cf *io, int fft_size);                                      /* fastddc.c:91-104 */
/* spectrum is NOT modified (the reference swaps it in place; we work on a copy) */
orc_dsa_status_t orc_fastddc_inv_cc(const orc_cf *spectrum, orc_cf *out, const orc_fastddc_t *ddc, const orc_cf *taps_fft, orc_dsa_status_t st); /* fastddc.c:106-166 */

/* ---- f2: the remaining simple blocks of the AM/SSB receive chains and the waterfall path ---- */
void  orc_amdemod_cf(const orc_cf *in, float *out, int n);                               /* libcsdr.c:861-873 */
void  orc_amdemod_estimator_cf(const orc_cf *in, float *out, int n, float alpha, float beta); /* :875-901 */
float orc_fmdemod_atan_cf(const orc_cf *in, float *out, int n, float last_phase);        /* :1004-1019 */
typedef struct { float last_input, last_output; } orc_dcblock_t;                         /* libcsdr.h:110-114 */
orc_dcblock_t orc_dcblock_ff(const float *in, float *out, int n, float a, orc_dcblock_t preserved); /* libcsdr.c:903-918 */
float orc_fastdcblock_ff(const float *in, float *out, int n, float last_dc_level);       /* :920-941 */
float orc_agc_ff(const float *in, float *out, int n, float reference, float attack_rate, float decay_rate, float max_gain,
                 short hang_time, short attack_wait_time, float gain_filter_alpha, float last_gain); /* libcsdr_gpl.c:163-260 */
void  orc_realpart_cf(const orc_cf *in, float *out, int n);                              /* csdr.c:634-645 */
void  orc_logpower_cf(const orc_cf *in, float *out, int n, float add_db);                /* libcsdr.c:1296-1303 */
void  orc_precalculate_window(float *windowt, int size, int window);                     /* :1256-1267 */
void  orc_apply_precalculated_window_c(const orc_cf *in, orc_cf *out, int size, const float *windowt); /* :1269-1276 */

/* ---- f3: IMA ADPCM codec -- integer, bit exact ---- */
typedef struct { int index, previousValue; } orc_adpcm_t;                                 /* ima_adpcm.h:5-8 */
orc_adpcm_t orc_encode_ima_adpcm_i16_u8(const short *in, unsigned char *out, int input_length, orc_adpcm_t state); /* ima_adpcm.c:154-163 */
orc_adpcm_t orc_decode_ima_adpcm_u8_i16(const unsigned char *in, short *out, int input_length, orc_adpcm_t state); /* :165-174 */
/* one block of `csdr compress_fft_adpcm_f_u8` (csdr.c:1745-1768): 10 pad values + fft_size dB values, x100 -> short, encoded from a zero state */
void orc_compress_fft_adpcm_f_u8(const float *in, unsigned char *out, int fft_size);

/* ---- whole-stream models of the csdr CLI loops (block framing included) ---- */
/* csdr.c:877-925: 1024-sample chunks, phase threaded through; n need not be a multiple of 1024 only for the last chunk */
float orc_stream_shift_addition_cc(const orc_cf *in, orc_cf *out, long n, float rate, float starting_phase, int chunk);
/* csdr.c:1160-1176: y[k] = sum_t h[t] x[D k + t], all k with D k + taps <= n; returns #outputs */
long  orc_stream_fir_decimate_cc(const orc_cf *in, orc_cf *out, long n, int decimation, const float *taps, int taps_length);
/* README.md:66 chain on one stream: u8 IQ -> shift -> fir_decimate -> fmdemod -> fractional_decimator(int rate) -> deemphasis -> s16.
 * audio_f (optional) receives the float audio before convert_f_s16.  Returns #audio samples. */
long  orc_stream_wfm_chain(const unsigned char *iq_u8, long n_complex, float shift_rate, int decimation,
                           const float *taps, int taps_length, int frac_rate, float tau, int audio_rate,
                           short *audio_s16, float *audio_f);

#ifdef __cplusplus
}
#endif
#endif
