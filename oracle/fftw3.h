/* oracle/fftw3.h -- TEST INFRASTRUCTURE, not product code.
 *
 * Minimal stand-in for the FFTW3 single-precision API, written from the public
 * FFTW3 interface documentation (FFTW is a third-party dependency of the
 * reference that is NOT vendored in /root/reference and is not installed in
 * this image: README.md:31 "libfftw3-dev", Makefile:39 "-lfftw3f").
 *
 * Only the seven entry points the reference's FFT shim calls are declared
 * (fft_fftw.c:9,19,29,38,43 and the fft_malloc/fft_free macros fft_fftw.h:11-12).
 * The implementation is oracle/fftw_shim.c (double-precision CPU FFT).
 */
#ifndef ORACLE_FFTW3_SHIM_H
#define ORACLE_FFTW3_SHIM_H
#include <stdio.h>   /* the reference's libcsdr.h uses FILE without including stdio */
#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef float fftwf_complex[2];
typedef struct oracle_fftwf_plan_s *fftwf_plan;

#define FFTW_FORWARD  (-1)
#define FFTW_BACKWARD (+1)
#define FFTW_MEASURE  (0U)
#define FFTW_ESTIMATE (1U << 6)

fftwf_plan fftwf_plan_dft_1d(int n, fftwf_complex *in, fftwf_complex *out, int sign, unsigned flags);
fftwf_plan fftwf_plan_dft_r2c_1d(int n, float *in, fftwf_complex *out, unsigned flags);
fftwf_plan fftwf_plan_dft_c2r_1d(int n, fftwf_complex *in, float *out, unsigned flags);
void fftwf_execute(const fftwf_plan p);
void fftwf_destroy_plan(fftwf_plan p);
void *fftwf_malloc(size_t n);
void fftwf_free(void *p);

#ifdef __cplusplus
}
#endif
#endif
